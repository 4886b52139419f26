"""Serial stand-in for caput.mpiutil."""
import numpy as np

rank = 0
size = 1
rank0 = True
world = None


def barrier():
    pass


def bcast(x, root=0):
    return x


def mpirange(*args):
    return list(range(*args))


def partition_list_mpi(lst):
    return list(lst)


def split_local(n):
    return np.array([n, 0, n])


def split_m(n, nchunk):
    base, rem = divmod(n, nchunk)
    num = np.array([base + (1 if i < rem else 0) for i in range(nchunk)])
    end = np.cumsum(num)
    start = end - num
    return np.array([num, start, end])


def split_all(n):
    return np.array([[n], [0], [n]])


def transpose_blocks(row_array, shape):
    return row_array


def allreduce(x, op=None):
    return x


class MPILogFilter(object):
    pass
