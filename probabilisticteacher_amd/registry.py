"""Name -> object registries: the plug-in surface the reference gets from detectron2.

The same registry names and registered class/function names are kept so that the reference's YAML strings
(`MODEL.META_ARCHITECTURE: "GuassianGeneralizedRCNN"`, `MODEL.BACKBONE.NAME: "build_vgg_backbone"`, ...)
select the MI355X-native implementations unchanged.
"""


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        if name in self._obj_map:
            raise KeyError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
