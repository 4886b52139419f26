def bit_truncate_max_complex(arr, rel, maxl):
    raise NotImplementedError("caput.truncate is not available in this container")
