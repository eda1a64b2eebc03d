"""Error types raised across the calculator boundary, same names and bases as the reference's
PyMaSC/core/exceptions.py:4-21 so the callers' ``except`` clauses keep working
(PyMaSC/pymasc.py:247-250 catches ReadUnsortedError per input file)."""


try:   # inside a PyMaSC installation: be catchable as the reference's own class (pymasc.py:247-250)
    from PyMaSC.core.exceptions import ReadUnsortedError as _RefReadUnsortedError
except Exception:   # stand-alone
    _RefReadUnsortedError = IndexError


class ReadUnsortedError(_RefReadUnsortedError):
    """Reads were not sorted by position, or a finished chromosome reappeared (mscc.pyx:351-364)."""


class ReadsTooFew(IndexError):
    pass


class InputUnseekable(Exception):
    pass


class NothingToCalc(Exception):
    pass
