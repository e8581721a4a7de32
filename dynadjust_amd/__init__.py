"""dynadjust_amd -- MI355X-native hot path of DynAdjust's dna_adjust (phased least squares).

Only what the path needs: the HIP kernels + C-ABI (csrc/, libdnagpu.so), the
ctypes binding (_lib, device) and the Python mirror of the dna_adjust interface
(adjust).  See DESIGN.md.
"""
__all__ = ["_lib", "device"]
