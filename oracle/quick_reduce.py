"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy restatement of the QuickReduce all-reduce.

What it follows.  The reference's QuickReduce (ROCm only) is `sgl-kernel/csrc/allreduce/quick_all_reduce.cuh`: a two-shot
all-reduce (`AllReduceTwoshot`, reduce-scatter then all-gather) whose transport goes through a codec --
`CodecFP` (:19-48), `CodecQ4` (:50-186), `CodecQ6` (:188-344), `CodecQ8` (:346-480) -- selected by
`QuickReduceRegime` (`python/sglang/srt/distributed/device_communicators/quick_all_reduce.py:47-52`).  The integer codecs
quantise blocks of 32 values with one half-precision scale:
    decoding_scale = absmax * (-1/R)            (kScaleFactor: -1/8 for Q4, -1/32 for Q6, -1/128 for Q8)
    encoding_scale = 1 / (decoding_scale + eps) (kScaleEpsilon, packed_rcp)
    w = clamp(x * encoding_scale, -R, R - 1)    (kRangeMin / kRangeMax);  q = rint(w) + R (kRangeBias)
    x' = (q - R) * decoding_scale
and every element is quantised twice on its way: once by the sending rank (reduce-scatter), once by the slice owner after
the sum (all-gather).  That kernel cannot be built or run in the build container (HIP device code, no GPU), and its wire
layout is private to it, so parity is pinned as follows:
  * the codec formulas above are restated here and in csrc/allreduce.hip (fp32 arithmetic with the scale rounded to half,
    where the reference uses packed half / bfloat16 arithmetic) and the HIP kernel must match THIS file bit for bit;
  * the reference's own acceptance test (`test/srt/test_quick_allreduce.py:131-165`: integer payloads in [1, 23],
    `atol = 1.25 * world_size`, `rtol = 0.5 * world_size` against the exact sum) is applied to both.
"parity unpinned" against the reference's kernel output itself (no way to execute it here); pinned against its codec
definition and its test's bound.
"""
import numpy as np

EPS = np.float32(2.0 ** -24)  # the smallest positive half (kScaleEpsilon 0x0001)
BITS = {1: 8, 2: 6, 3: 4}     # QuickReduceRegime.INT8 / INT6 / INT4


def _encode(x: np.ndarray, bits: int):
    """x fp32 [n], n % 32 == 0 -> (q uint8 [n] in [0, 2R), dec half [n/32])."""
    R = np.float32(1 << (bits - 1))
    blocks = x.reshape(-1, 32)
    am = np.abs(blocks).max(axis=1).astype(np.float32)
    dec_h = (am * np.float32(-1.0) / R).astype(np.float32).astype(np.float16)
    dec = dec_h.astype(np.float32)
    with np.errstate(divide="ignore"):
        enc = (np.float32(1.0) / (dec + EPS)).astype(np.float32)
    w = np.rint((blocks * enc[:, None]).astype(np.float32))
    q = (np.clip(w, -R, R - np.float32(1.0)) + R).astype(np.int32)
    return q.reshape(-1), dec_h


def _decode(q: np.ndarray, dec_h: np.ndarray, bits: int) -> np.ndarray:
    R = 1 << (bits - 1)
    return ((q.reshape(-1, 32) - R).astype(np.float32) * dec_h.astype(np.float32)[:, None]).astype(np.float32).reshape(-1)


def quick_all_reduce(parts, regime: int, round_out, max_bytes: int = 16 * 1024 * 1024):
    """parts: list of W arrays (fp32 views of the ranks' 16-bit inputs, same length n, n % 32 == 0).
    regime 1 / 2 / 3.  round_out: fp32 array -> fp32 array rounded to the 16-bit output dtype.
    max_bytes: the communicator's staging capacity (decides the chunking and therefore the slice boundaries).
    Returns the fp32 view of the (identical on every rank) result."""
    bits = BITS[regime]
    W = len(parts)
    n = parts[0].size
    assert n % 32 == 0
    cap_units = (max_bytes - 64) * 2 // (2 * bits + 1)
    cap_units &= ~(4 * 8 - 1)
    out = np.empty(n, dtype=np.float32)
    total_units = n // 8
    u0 = 0
    while u0 < total_units:
        nu = min(total_units - u0, cap_units)
        lo_v, hi_v = u0 * 8, (u0 + nu) * 8
        enc = [_encode(p[lo_v:hi_v].astype(np.float32), bits) for p in parts]        # phase A on every rank
        per = -(-nu // W)
        per = (per + 3) & ~3
        res = np.empty(nu * 8, dtype=np.float32)
        for r in range(W):                                                             # phase B on the owner r
            lo, hi = min(r * per, nu) * 8, min(r * per + per, nu) * 8
            if lo >= hi:
                continue
            acc = np.zeros(hi - lo, dtype=np.float32)
            for p in range(W):                                                         # rank order, fp32
                q, d = enc[p]
                acc = (acc + _decode(q[lo:hi], d[lo // 32:hi // 32], bits)).astype(np.float32)
            q2, d2 = _encode(acc, bits)
            # phase C: everyone decodes these bytes (into a zeroed accumulator: a zero comes out as +0, whatever the scale's sign)
            res[lo:hi] = (np.zeros(hi - lo, dtype=np.float32) + _decode(q2, d2, bits)).astype(np.float32)
        out[lo_v:hi_v] = round_out(res)
        u0 += nu
    return out
