"""Writer process of the product files (storage.py, DRIFTMI_IO_PROCS): reads length-prefixed pickled tasks
(temporary file name, dataset specifications pointing at POSIX shared-memory blocks, attributes) from stdin, writes the
HDF5 file with its OWN libdriftio / libhdf5, answers ("ok", bytes written) or ("err", text) on stdout.  Host only: never
touches the GPU."""
import pickle
import struct
import sys


def main():
    from driftscan_amd import storage

    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr            # nothing but the protocol may go to the pipe
    while True:
        hdr = inp.read(8)
        if len(hdr) < 8:
            return
        task = pickle.loads(inp.read(struct.unpack("<Q", hdr)[0]))
        try:
            res = ("ok", storage._proc_task(*task))
        except Exception as e:   # reported to the writer thread, which raises it there
            res = ("err", repr(e))
        blob = pickle.dumps(res)
        out.write(struct.pack("<Q", len(blob)))
        out.write(blob)
        out.flush()


if __name__ == "__main__":
    main()
