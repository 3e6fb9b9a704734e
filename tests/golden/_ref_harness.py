"""What the fixture generators need to IMPORT /root/reference's Python modules in the authoring container (CPU only).

The reference's modules import, at module level, third-party packages this image lacks (pytorch3d, plyfile, simple_knn,
diff_gaussian_rasterization, cv2, pyrr, pyvista, pyvistaqt, dreifus, wandb, tensorboard).  `install_placeholders()` makes
those names importable as EMPTY modules whose every attribute raises when called: the generators only run reference
methods that never reach them (a call aborts the generation).  Nothing of the reference is edited, wrapped or copied.
"""
import os
import sys
import types

REF = "/root/reference"
ABSENT = ("pytorch3d", "pytorch3d.ops", "pytorch3d.transforms", "plyfile", "simple_knn", "simple_knn._C",
          "diff_gaussian_rasterization", "cv2", "pyrr", "pyvista", "pyvistaqt", "dreifus", "dreifus.pyvista", "wandb",
          "tensorboard", "torch.utils.tensorboard")


class _Absent(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"

        def absent(*a, **k):
            raise RuntimeError(f"{full} is absent in this image and must not be reached by the pinned methods")
        absent.__name__ = name
        return absent


def install_placeholders():
    placed = []
    for name in ABSENT:
        try:
            __import__(name)
        except Exception:
            mod = _Absent(name)
            mod.__path__ = []
            sys.modules[name] = mod
            placed.append(name)
    return placed


def enter_reference():
    """sys.path -> the reference; its `scene` and `loss` packages are entered WITHOUT running their __init__.py (which pull in the
    dataset readers): submodules are then imported from their own files, unedited."""
    assert os.path.isdir(REF), "needs /root/reference (authoring container only)"
    placed = install_placeholders()
    sys.path.insert(0, REF)
    for name in ("scene", "loss"):       # (loss/__init__.py pulls in the evaluation data readers the same way)
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF, name)]
        sys.modules[name] = pkg
    return placed
