"""nas_3d_unet_amd: MI355X (gfx950) native hot path of the NAS-3D-U-Net project.

Python host mirroring the reference operator API (prim_ops / cell / nas / searched / loss /
genotype) over the libn3d HIP kernels (csrc/, include/n3d.h).  Importing the package needs no
GPU; running any op does, and fails loudly without the built library or a gfx950 device.
"""
from . import _lib  # noqa: F401

__all__ = ["prim_ops", "cell", "nas", "searched", "loss", "genotype"]
