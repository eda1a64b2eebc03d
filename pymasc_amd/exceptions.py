"""Error types raised across the calculator boundary, same names and bases as the reference's
PyMaSC/core/exceptions.py:4-21 so the callers' ``except`` clauses keep working
(PyMaSC/pymasc.py:247-250 catches ReadUnsortedError per input file).  Inside a PyMaSC installation they are
subclasses of the reference's own classes (like ``pymasc_amd.result`` binds the reference's result types)."""


try:   # inside a PyMaSC installation: be catchable as the reference's own classes
    from PyMaSC.core import exceptions as _ref
    _bases = (_ref.ReadUnsortedError, _ref.ReadsTooFew, _ref.InputUnseekable, _ref.NothingToCalc)
except Exception:   # stand-alone: the reference's bases (core/exceptions.py:4,9,14,19)
    _bases = (IndexError, IndexError, Exception, Exception)


class ReadUnsortedError(_bases[0]):
    """Reads were not sorted by position, or a finished chromosome reappeared (mscc.pyx:351-364)."""


class ReadsTooFew(_bases[1]):
    pass


class InputUnseekable(_bases[2]):
    pass


class NothingToCalc(_bases[3]):
    pass
