"""CPU restatement of the bit truncation applied to the beam-transfer blocks before they are written
(drift/core/beamtransfer.py:641-646 -> caput.truncate.bit_truncate_max_complex).

TEST INFRASTRUCTURE — never imported by the product.

**Parity unpinned.**  `caput` is a third-party dependency that is not vendored with the reference
(`caput @ git+https://github.com/radiocosmology/caput.git`, unpinned, pyproject.toml:26) and is not
installed here; the reference's tests hold no vectors for it.  What is restated is its documented
contract — "truncate using a relative per element and per the maximum of the last dimension": every
real and imaginary part of `val[i, j]` is rounded so that its error stays below
`max(prec * |val[i, j]|, prec_max_row * max_j |val[i, j]|)` with as many trailing mantissa bits zero as
that bound allows — with one concrete rounding rule, stated here and implemented identically (same IEEE
operations in the same order) by the HIP kernel `dm_bit_truncate_max_complex`:

    x = (+-) man * 2^e  (man the 53-bit integer mantissa);  errm = floor(err / 2^e);  unchanged if errm < 1
    k = floor(log2(errm));  man is rounded to the nearest multiple of 2^(k+1), ties to even

so |result - x| <= 2^k * 2^e <= err.  Whether caput rounds to 2^(k+1) or 2^k is not verifiable here; both
satisfy the contract and any such file is read the same way.
"""
import numpy as np


def bit_truncate_f64(x, err):
    x = np.ascontiguousarray(x, dtype=np.float64)
    err = np.broadcast_to(np.asarray(err, dtype=np.float64), x.shape)
    bits = x.view(np.uint64)
    ex = ((bits >> np.uint64(52)) & np.uint64(0x7FF)).astype(np.int64)
    man = bits & np.uint64(0x000FFFFFFFFFFFFF)
    e = np.where(ex == 0, -1074, ex - 1075)
    man = np.where(ex == 0, man, man | (np.uint64(1) << np.uint64(52)))
    with np.errstate(over="ignore", invalid="ignore"):
        errs = np.ldexp(err, (-e).astype(np.int64))
    active = (err > 0.0) & (x != 0.0) & (ex != 0x7FF) & (errs >= 1.0)
    errs_c = np.where(active, np.minimum(errs, 4611686018427387904.0), 1.0)
    errm = np.where(errs_c >= 4611686018427387904.0, np.uint64(1) << np.uint64(62), errs_c.astype(np.uint64))
    # k = position of the highest set bit
    k = np.zeros(x.shape, dtype=np.uint64)
    t = errm.copy()
    for s in (32, 16, 8, 4, 2, 1):
        big = t >= (np.uint64(1) << np.uint64(s))
        k = np.where(big, k + np.uint64(s), k)
        t = np.where(big, t >> np.uint64(s), t)
    q = np.uint64(1) << (k + np.uint64(1))
    half = np.uint64(1) << k
    r = man & (q - np.uint64(1))
    base = man - r
    up = (r > half) | ((r == half) & (((base >> (k + np.uint64(1))) & np.uint64(1)) == 1))
    base = np.where(up, base + q, base)
    with np.errstate(over="ignore", invalid="ignore"):
        out = np.ldexp(base.astype(np.float64), e.astype(np.int64))
    out = np.where(bits >> np.uint64(63) == 1, -out, out)
    return np.where(active, out, x)


def bit_truncate_max_complex(val, prec, prec_max_row):
    """val: (nrows, ncols) complex128.  Returns the truncated copy."""
    val = np.asarray(val, dtype=np.complex128)
    re, im = np.ascontiguousarray(val.real), np.ascontiguousarray(val.imag)
    with np.errstate(over="ignore", invalid="ignore"):
        abs2 = re * re + im * im
        # NaNs do not take part in the maxima (C fmax, as the kernel's reductions)
        row_max = np.fmax(np.fmax.reduce(abs2, axis=-1, keepdims=True), 0.0) if val.shape[-1] else 0.0
        floor_err = prec_max_row * np.sqrt(row_max)
        err = np.fmax(prec * np.sqrt(abs2), floor_err)
    out = np.empty(val.shape, dtype=np.complex128)   # not re + 1j * im: that turns (inf, nan) into (nan, nan)
    out.real = bit_truncate_f64(re, err)
    out.imag = bit_truncate_f64(im, err)
    return out
