"""Host orchestration of beam-transfer generation on the GPU.

Groups the requested (frequency, baseline) pairs by the HEALPix resolution the
reference would pick for them (``nside_for_lmax`` of the per-baseline band limit,
drift/core/telescope.py:1179-1184, :1288), evaluates the beams, synthesises the
visibility response maps and transforms them straight into the m-ordered
``beam_m`` blocks (``dm_bt_beam_cyl`` / ``dm_bt_maps`` / ``dm_bt_sht``).
"""
import os

import numpy as np

from . import healpix
from .device import get_context


def telescope_frame(zenith):
    """xhat (East), yhat (North), zhat (up) in sky cartesian coordinates — the unrotated
    cylinder frame of cylbeam.py:129 (phihat, -thetahat, zenith)."""
    t, p = zenith
    that = np.array([np.cos(t) * np.cos(p), np.cos(t) * np.sin(p), -np.sin(t)])
    phat = np.array([-np.sin(p), np.cos(p), 0.0])
    zhat = np.array([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)])
    return np.concatenate([phat, -that, zhat])


def _nside_of(tel, lmax_bf):
    boost = tel.accuracy_boost if tel.num_pol_sky == 1 else 1
    return np.array([healpix.nside_for_lmax(int(l), boost) for l in lmax_bf], dtype=np.int64)


def fill_beam_m(tel, beam_m, f_list, b_list, row_f=None, row_b=None, F=None, B=None, mmax=None,
                max_bytes=6 << 30, ctx=None, m_range=None):
    """Fill rows of the device array ``beam_m`` (mmax+1, F, 2, B, P, lside+1) for the
    (frequency, baseline) pairs (f_list[i], b_list[i]); the destination row of pair i is
    (row_f[i], row_b[i]) (defaults: the telescope indices themselves).  With ``m_range = (m_lo, m_hi)``
    only those m-blocks are transformed and ``beam_m`` holds m_hi - m_lo + 1 blocks."""
    ctx = ctx or get_context()
    f_list = np.asarray(f_list, dtype=np.int64).reshape(-1)
    b_list = np.asarray(b_list, dtype=np.int64).reshape(-1)
    row_f = f_list if row_f is None else np.asarray(row_f, dtype=np.int64).reshape(-1)
    row_b = b_list if row_b is None else np.asarray(row_b, dtype=np.int64).reshape(-1)
    F = tel.nfreq if F is None else F
    B = tel.nbase if B is None else B
    mmax = tel.mmax if mmax is None else mmax
    lside = tel.lmax
    pol = tel.num_pol_sky > 1
    P = 4 if pol else 1
    frame = telescope_frame(tel.zenith)
    # the band limits and resolutions of the columns depend on the telescope and the column list only: remembered per list
    # (a rank calls this once per m-range with the same columns; the host work between two calls is idle GPU time)
    # (the key carries everything the band limits and resolutions follow from: a telescope changed after a first call
    # gets new ones; `TransitTelescope.__getstate__` keeps the memo out of the telescope pickle)
    memo = tel.__dict__.setdefault("_btgen_memo", {})
    mkey = (f_list.tobytes(), b_list.tobytes(), row_f.tobytes(), row_b.tobytes(), lside, float(tel.l_boost), float(tel.accuracy_boost), int(tel.num_pol_sky),
            np.asarray(tel.wavelengths, dtype=np.float64).tobytes(), np.asarray(tel.baselines, dtype=np.float64).tobytes(),
            float(getattr(tel, "u_width", 0.0)), float(getattr(tel, "v_width", 0.0)),
            np.asarray(tel.beamclass).tobytes(), np.asarray(tel.uniquepairs).tobytes())
    if mkey not in memo:
        if len(memo) > 8:
            memo.clear()
        lmax_bf, _ = tel.baseline_lmax(b_list, f_list)
        if (lmax_bf > lside).any():
            raise ValueError("a baseline's natural lmax exceeds the telescope lmax (force_lmax too small)")
        memo[mkey] = (lmax_bf, _nside_of(tel, lmax_bf))
    lmax_bf, nsides = memo[mkey]
    cmemo = tel.__dict__.setdefault("_btgen_chunk_memo", {})
    pairs = tel.uniquepairs
    cls = np.asarray(tel.beamclass)
    wl = tel.wavelengths
    # healpy.map2alm knobs the reference reaches through cora (telescope.py:1179-1191, :1288-1312): `iter` (healpy's
    # documented default 3 is the telescope's default) and ring weights
    niter = int(getattr(tel, "sht_iter", 0) or 0)
    ringw = getattr(tel, "sht_ring_weights", None)   # None, or {nside: (4 nside - 1) factors}

    for nside in np.unique(nsides):
        sel = np.nonzero(nsides == nside)[0]
        npix = healpix.npix(int(nside))
        nring = 4 * int(nside) - 1
        cth, sth = healpix.ring_trig(int(nside))
        # chunk columns so that maps + ring-DFT output + twiddles stay within the budget
        lgrp = int(lmax_bf[sel].max())
        mtop = min(mmax, lgrp)
        nmr = (2 * mtop + 1) if m_range is None else 2 * max(min(m_range[1], mtop) - m_range[0] + 1, 1)
        # default: the fused path (dm_bt_columns / dm_bt_columns_iter) — the Stokes maps are never written, and healpy's
        # `iter` refinements run in harmonic space (dm_bt_columns_iter); DRIFTMI_BT_MAPS=1 forces the two-call path
        # (maps materialised, dm_bt_sht_opts) for comparisons
        # (complex field patterns — a class that declares `complex_beams = True`, or whose beam() returns complex maps —
        # go through the same fused path: the kernels form _construct_pol_complex themselves)
        fused = os.environ.get("DRIFTMI_BT_MAPS") != "1"
        pixel_refine = niter and not fused and os.environ.get("DM_SHT_PIXEL_REFINE") == "1"
        priv = 0
        if pixel_refine:   # residual maps, all m of the group's columns in G and in the private coefficient buffer
            nmr = 2 * (lgrp + 1)
            priv = 2 * (lgrp + 1) * (lside + 1)
        elif niter:
            # private coefficient buffers of the refinement (sum, increment, product): the caller's m, or 0 .. max(m_hi, mcut)
            # when the range reaches into the m the polar rings couple
            from ._lib import bt_alias_info
            nalias, mcut = bt_alias_info(int(nside), cth, sth, pol, lgrp)
            e_lo, e_hi = (0, mmax) if m_range is None else (int(m_range[0]), int(m_range[1]))
            pad = 0
            if mcut >= 0 and e_lo <= mcut:
                e_lo, e_hi, pad = 0, max(e_hi, mcut), 2 * nalias
            ncnt = max(min(e_hi, lgrp) - e_lo + 1, 1)
            nmr = 2 * ncnt
            # three buffers of (m, 2, P, coefficients + alias slots) per column + the alias-ring synthesis
            priv = 3 * 2 * (e_hi - e_lo + 1) * (lgrp + 1 + pad) + (2 * (mcut + 1) * pad if pad else 0)
        per_col = P * 16 * ((0 if fused else (2 if pixel_refine else 1)) * npix + nmr * nring + priv)
        fixed = 0 if fused else nmr * npix * 16
        ncol_max = max(1, int((max_bytes - fixed) // per_col)) if max_bytes > fixed else 1
        # keep all baselines of a frequency together and in order: dm_bt_sht merges such runs
        order = sel[np.lexsort((row_b[sel], row_f[sel]))]
        for c0 in range(0, order.size, ncol_max):
            cols = order[c0 : c0 + ncol_max]
            # distinct (frequency, beam class) beams of this chunk and the per-column constants: they follow from the column
            # list alone and are remembered with it (27 648 columns per chunk at configs[2]: the Python loop below is
            # tens of milliseconds of idle GPU per call otherwise)
            ckey = (mkey, int(nside), int(c0), int(ncol_max))
            if ckey not in cmemo:
                if len(cmemo) > 64:
                    cmemo.clear()
                keys = {}
                feed_of = {}
                bi = np.empty(cols.size, dtype=np.int32)
                bj = np.empty(cols.size, dtype=np.int32)
                for k, c in enumerate(cols):
                    fi_, fj_ = pairs[b_list[c]]
                    for which, feed in ((bi, fi_), (bj, fj_)):
                        key = (int(f_list[c]), int(cls[feed]))
                        if key not in keys:
                            keys[key] = len(keys)
                            feed_of[key] = int(feed)
                        which[k] = keys[key]
                cmemo[ckey] = (keys, feed_of, bi, bj, np.ascontiguousarray(tel.baselines[b_list[cols]] / wl[f_list[cols]][:, None]),
                               np.ascontiguousarray(row_f[cols]), np.ascontiguousarray(row_b[cols]),
                               np.ascontiguousarray(lmax_bf[cols]), int(lmax_bf[cols].max()))
            keys, feed_of, bi, bj, uv, rf_c, rb_c, lm_c, lm_top = cmemo[ckey]
            ncomp = 2 if pol else 1
            # the reference's plug-in interface: beam(feed, freq) evaluated by the telescope class on the host
            # (telescope.py:954-973 keys the maps by beam class as here), uploaded once
            specs = {key: tel.beam_spec(key[1], key[0]) for key in keys}
            hostb = {key: tel._beam_host(feed_of[key], key[0], int(nside)) for key in keys if specs[key] is None}
            cbeams = any(np.iscomplexobj(b) for b in hostb.values())   # complex patterns: _construct_pol_complex
            beams = ctx.empty((len(keys), npix * ncomp), np.complex128 if cbeams else np.float64)
            dev_specs, dev_rows = [], []
            for key, idx in keys.items():
                if specs[key] is None:
                    beams[idx].copy_(ctx.to_device(hostb[key].reshape(-1)))
                    continue
                kind, tab, fwhm_ns = specs[key]
                if cbeams:   # a device-evaluated (real) pattern next to complex ones
                    tmp = ctx.empty((npix * ncomp,), np.float64)
                    ctx.bt_beam_cyl(int(nside), cth, sth, frame, kind, tab, fwhm_ns, tmp)
                    beams[idx].copy_(tmp)
                else:
                    dev_specs.append(specs[key])
                    dev_rows.append(idx)
            # all device-evaluated patterns of the chunk in one call (geometry and tables staged once)
            ctx.bt_beams_cyl(int(nside), cth, sth, frame, dev_specs, beams, dev_rows)
            del hostb
            if fused:
                # (with the refinement every chunk of the group is transformed against the GROUP's band limit: the set of
                # aliased (ring, m) pairs follows from it, and the bits of a block must not depend on how the columns were
                # chunked — the chunking depends on the m-range of the caller)
                ctx.bt_columns(int(nside), cth, sth, frame, pol, beams, uv, bi, bj, lside, mmax,
                               lgrp if niter else lm_top, F, B,
                               rf_c, rb_c, lm_c, beam_m, m_range=m_range,
                               ring_w=None if ringw is None else ringw.get(int(nside)), niter=niter)
                del beams
                continue
            maps = ctx.empty((cols.size, P, npix), np.complex128)
            ctx.bt_maps(int(nside), cth, sth, frame, pol, beams, uv, bi, bj, maps)
            ctx.bt_sht(int(nside), cth, sth, pol, lside, mmax, lgrp if niter else lm_top, F, B, rf_c,
                       rb_c, lm_c, maps, beam_m, m_range=m_range, niter=niter,
                       ring_w=None if ringw is None else ringw.get(int(nside)))
            del maps, beams
    return beam_m


def beam_m_all(tel, ctx=None, max_bytes=6 << 30, m_range=None):
    """Device tensor (mmax+1, F, 2, B, P, L) of every m-block of the telescope (zeros for
    skipped frequencies / baselines, like the reference's beam_m accessor), or of the blocks
    m_range = (m_lo, m_hi) only."""
    ctx = ctx or get_context()
    F, B, P, L = tel.nfreq, tel.nbase, tel.num_pol_sky, tel.lmax + 1
    nm = tel.mmax + 1 if m_range is None else m_range[1] - m_range[0] + 1
    beam_m = ctx.zeros((nm, F, 2, B, P, L), np.complex128)
    ff, bb = np.meshgrid(tel.included_freq, tel.included_baseline, indexing="ij")
    fill_beam_m(tel, beam_m, ff.ravel(), bb.ravel(), ctx=ctx, max_bytes=max_bytes, m_range=m_range)
    npol_inc = len(tel.included_pol)
    if npol_inc < P:
        beam_m[:, :, :, :, npol_inc:, :] = 0  # skip_pol / skip_V leave zero entries (telescope.py:1298-1314)
    return beam_m


def transfer_matrices(tel, bl, fi, global_lmax=True, ctx=None):
    """numpy (nfb, P, lside+1, 2*lside+1): the reference's ``transfer_matrices`` layout,
    rebuilt from the m-ordered device result (m >= 0 in column m, m < 0 wrapped to the end)."""
    ctx = ctx or get_context()
    lside = tel.lmax
    P, L = tel.num_pol_sky, lside + 1
    nfb = bl.size
    bm = ctx.zeros((lside + 1, nfb, 2, 1, P, L), np.complex128)
    fill_beam_m(tel, bm, fi, bl, row_f=np.arange(nfb), row_b=np.zeros(nfb, dtype=np.int64), F=nfb, B=1, mmax=lside,
                ctx=ctx)
    h = bm.cpu().numpy()  # (m, nfb, 2, 1, P, L)
    out = np.zeros((nfb, P, L, 2 * lside + 1), dtype=np.complex128)
    for m in range(lside + 1):
        out[:, :, :, m] = h[m, :, 0, 0]
        if m > 0:
            out[:, :, :, -m] = (-1) ** m * h[m, :, 1, 0].conj()
    return out
