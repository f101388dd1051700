"""gst_visdial_amd -- MI355X-native hot path of gst-visdial (enc_dec_a train/eval step).

Python host code over a C ABI (include/gstvd_hip.h -> gst_visdial_amd/lib/libgstvd_hip.so) of
hand-written gfx950 HIP kernels.  No CPU fallback: importing is cheap, using any op without the
built library or without a GPU raises.
"""
__version__ = "0.1.0"
