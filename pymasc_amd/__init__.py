"""pymasc_amd -- MI355X-native per-chromosome strand cross-correlation (PyMaSC BitArray path).

Host-side mirror of the reference's calculator interface over a C-ABI HIP library:
  ffi.Context            ctypes binding of include/pymasc_amd.h
  calculator.CCHipCalculator   drop-in for PyMaSC.core.bitarray.mscc.CCBitArrayCalculator
  result                 NCCResult / MSCCResult / ... value-equal to PyMaSC.result
"""
__version__ = "0.1.0"
