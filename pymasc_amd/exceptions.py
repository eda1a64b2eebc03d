"""Error types raised across the calculator boundary, same names and bases as the reference's
PyMaSC/core/exceptions.py:4-21 so the callers' ``except`` clauses keep working
(PyMaSC/pymasc.py:247-250 catches ReadUnsortedError per input file)."""


class ReadUnsortedError(IndexError):
    """Reads were not sorted by position, or a finished chromosome reappeared (mscc.pyx:351-364)."""


class ReadsTooFew(IndexError):
    pass


class InputUnseekable(Exception):
    pass


class NothingToCalc(Exception):
    pass
