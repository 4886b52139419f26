"""Stand-in for mpi4py (absent from this image), single process: just enough for
`from mpi4py import MPI` in drift/core/psestimation.py to import.  TEST INFRASTRUCTURE."""


class _Comm(object):
    def Allgatherv(self, *a, **k):
        return None


class _MPI(object):
    SUM = "sum"
    IN_PLACE = None
    DOUBLE = "double"
    COMM_WORLD = _Comm()


MPI = _MPI()
