"""MI355X-native video-captioning Transformer training / greedy-decode path.

Python host code (mirroring the reference's module API) over libvct_hip.so: hand-written gfx950
kernels behind the C ABI in include/vct_hip.h.  Import as `vct_amd` (see vct_amd/__init__.py)."""
__version__ = "0.1.0"
