"""handobjectconsist_amd -- MI355X-native render + photometric-warp hot path of
hassony2/handobjectconsist (meshreg/neurender + meshreg/warping), behind the reference's own
Python call signatures.  Native code: libmeshraster_hip.so (include/meshraster_hip.h)."""

__version__ = "0.1.0"
