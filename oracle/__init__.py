"""CPU oracle for the ProbabilisticTeacher teacher+student train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``probabilisticteacher_amd/`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker (never as the thing measured
or shipped).

Layout
------
``oracle/d2.py``     restatement of the detectron2==0.5 / torchvision==0.8.2 /
                     fvcore primitives the reference leans on.
                     Their source is NOT under /root/reference and is not
                     importable in this image  =>  **parity unpinned** for these
                     (guarded by brute-force / property tests only).
``oracle/pt.py``     restatement of the reference's own arithmetic
                     (/root/reference/pt/**), function by function, each citing
                     the file:line it follows.  **Pinned** by golden vectors in
                     ``tests/golden/*.npz`` that ``tools/gen_golden.py`` produced
                     by importing the real reference modules (on top of
                     ``oracle/d2.py`` standing in for detectron2) in the dev
                     container.
``oracle/csrc``      plain-C restatement of the integer / gather-heavy pieces
                     (greedy NMS, ROIAlign fwd/bwd) so the oracle finishes in
                     seconds; built by ``oracle/build.py`` with gcc.

All floating point is fp32 on the CPU (stock torch ops), matching the reference,
which is effectively fp32-only (SURVEY.md §2.2 AMP note).
"""
