"""Dataset registry and the Pascal-VOC directory reader (reference pt/data/datasets/builtin.py:118-149: every dataset of the
shipped configs is a VOC-format directory registered under a name; detectron2's `register_pascal_voc` /
`load_voc_instances` do the reading there).

Host side of the boundary: XML parsing and image decoding (Pillow) happen here; everything after the decoded uint8 tensor is
the device pipeline of data/mapper.py."""
import os
import xml.etree.ElementTree as ET
from typing import Callable, Dict, List, Sequence

import numpy as np
import torch

_DATASETS: Dict[str, Callable[[], List[dict]]] = {}
_METADATA: Dict[str, dict] = {}

CLASS_NAMES = {        # builtin.py:134-147
    1: ("car",),
    7: ("truck", "car", "rider", "person", "motorcycle", "bicycle", "bus"),
    8: ("truck", "car", "rider", "person", "train", "motorcycle", "bicycle", "bus"),
    20: ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse",
         "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"),
}
VOC_SPLITS = [         # builtin.py:120-130: (name, directory under the dataset root, split, number of classes)
    ("VOC2007_citytrain", "data/VOC2007_citytrain", "train", 8), ("VOC2007_foggytrain", "data/VOC2007_foggytrain", "train", 8),
    ("VOC2007_foggyval", "data/VOC2007_foggyval", "val", 8), ("VOC2007_citytrain1", "data/VOC2007_citytrain1", "train", 1),
    ("VOC2007_cityval1", "data/VOC2007_cityval1", "val", 1), ("VOC2007_bddtrain", "data/VOC2007_bddtrain", "train", 8),
    ("VOC2007_bddval", "data/VOC2007_bddval", "val", 8), ("VOC2007_kitti1", "data/kitti", "train", 1),
    ("VOC2007_sim1", "data/sim", "train", 1),
]


def load_voc_instances(dirname: str, split: str, class_names: Sequence[str]) -> List[dict]:
    """D2 load_voc_instances: ImageSets/Main/<split>.txt lists the ids; Annotations/<id>.xml, JPEGImages/<id>.jpg.  The XML's
    1-based inclusive corners become 0-based boxes by subtracting 1 from xmin / ymin only (D2's convention, which the
    evaluator undoes)."""
    with open(os.path.join(dirname, "ImageSets", "Main", split + ".txt")) as f:
        ids = [line.strip() for line in f if line.strip()]
    dicts = []
    for fid in ids:
        tree = ET.parse(os.path.join(dirname, "Annotations", fid + ".xml"))
        rec = {"file_name": os.path.join(dirname, "JPEGImages", fid + ".jpg"), "image_id": fid,
               "height": int(tree.findall("./size/height")[0].text), "width": int(tree.findall("./size/width")[0].text)}
        ann = []
        for obj in tree.findall("object"):
            name = obj.find("name").text
            if name not in class_names:
                continue
            bb = obj.find("bndbox")
            box = [float(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]
            box[0] -= 1.0
            box[1] -= 1.0
            diff = obj.find("difficult")
            ann.append({"category_id": class_names.index(name), "bbox": box, "difficult": int(diff.text) if diff is not None else 0})
        rec["annotations"] = ann
        dicts.append(rec)
    return dicts


def register_pascal_voc(name: str, dirname: str, split: str, class_names: Sequence[str], year: int = 2012) -> None:
    _DATASETS[name] = lambda: load_voc_instances(dirname, split, class_names)
    _METADATA[name] = {"thing_classes": list(class_names), "dirname": dirname, "split": split, "year": year,
                       "evaluator_type": "pascal_voc"}


def register_all_pascal_voc(root: str) -> None:
    for name, d, split, ncls in VOC_SPLITS:
        register_pascal_voc(name, os.path.join(root, d), split, CLASS_NAMES[ncls])


def get_dataset_dicts(names: Sequence[str], filter_empty: bool = False) -> List[dict]:
    """D2 get_detection_dataset_dicts: concatenation of the named datasets, optionally without annotation-less images"""
    out = []
    for n in names:
        if n not in _DATASETS:
            raise KeyError(f"dataset {n!r} is not registered (register_pascal_voc / DETECTRON2_DATASETS)")
        d = _DATASETS[n]()
        assert len(d), f"dataset {n!r} is empty"
        out += d
    if filter_empty:
        out = [d for d in out if len(d.get("annotations", []))]
    return out


def metadata(name: str) -> dict:
    return _METADATA[name]


def read_image(file_name: str, fmt: str = "BGR") -> torch.Tensor:
    """detection_utils.read_image + the mapper's HWC -> CHW transpose (dataset_mapper.py:162-169): uint8 (3, H, W)"""
    from PIL import Image, ImageOps
    with Image.open(file_name) as im:
        im = ImageOps.exif_transpose(im)          # D2 _apply_exif_orientation: rotated JPEGs are turned upright
        arr = np.asarray(im.convert("RGB"))
    if fmt == "BGR":
        arr = arr[:, :, ::-1]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


def to_mapper_input(d: dict, fmt: str = "BGR") -> dict:
    """dataset dict -> what DeviceTwoCropMapper takes: decoded image + boxes / classes (+ difficult flags)"""
    out = {"image": read_image(d["file_name"], fmt), "image_id": d["image_id"], "file_name": d["file_name"]}
    ann = d.get("annotations")
    if ann is not None:
        out["boxes"] = torch.tensor([a["bbox"] for a in ann], dtype=torch.float32).reshape(-1, 4)
        out["classes"] = torch.tensor([a["category_id"] for a in ann], dtype=torch.int64)
        out["difficult"] = torch.tensor([a.get("difficult", 0) for a in ann], dtype=torch.bool)
    return out


register_all_pascal_voc(os.getenv("DETECTRON2_DATASETS", ""))      # builtin.py:152-154
