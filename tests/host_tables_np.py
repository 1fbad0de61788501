"""Independent NumPy readings of the HOST-ONLY rows of the decode path.

The product's host front-end (libjxl_amd/csrc/host/jxh_*.h) and the oracle (oracle/jxlo_*.h) are near twins for these
rows, so holding one against the other proves little. What is here was written from the REFERENCE'S TEXT and shares no
code with either:
  * the dequantisation-table generator: lib/jxl/quant_weights.cc:47-357 (GetQuantWeightsDCT2 / Identity, Interpolate,
    Mult, GetQuantWeights, ComputeQuantTable for every mode) with lib/jxl/base/fast_math-inl.h:48-92 (FastLog2f,
    FastPow2f, FastPowf) and rational_polynomial-inl.h:60-97 evaluated step by step in float32;
  * the wire format of a coded DequantMatrices section: quant_weights.cc:373-511 (mode in 3 bits, F16 parameters, the x64
    scaling of seeds and weights), written here by a bit writer of its own (dec_bit_reader.h: LSB first);
  * the natural coefficient order: lib/jxl/ac_strategy.cc:28-79 (CoeffOrderAndLut, with CoefficientLayout of
    coeff_order_fwd.h:39-44);
  * the zero-density context: lib/jxl/ac_context.h:63-84 with its two tables taken from the reference's source
    (tests/golden/ref_constant_tables.json).
Test infrastructure, like the oracle: only tests/ may import it."""
import numpy as np

F = np.float32

# quant table kinds in the order of the enum (quant_weights.h:332-352) and the block counts DequantMatrices::required_size_x /
# _y list for them (quant_weights.h:367-370); ComputeQuantTable makes wrows = 8 * required_size_x, wcols = 8 * required_size_y
REQ_X = [1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16]
REQ_Y = [1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32]
MODE_LIBRARY, MODE_ID, MODE_DCT2, MODE_DCT4, MODE_DCT4X8, MODE_AFV, MODE_DCT, MODE_RAW = range(8)
# the mode each library table is defined with (quant_weights.cc:533-1106: DCT(), IDENTITY(), DCT2X2(), DCT4X4(), ...)
LIBRARY_MODE = [MODE_DCT, MODE_ID, MODE_DCT2, MODE_DCT4, MODE_DCT, MODE_DCT, MODE_DCT, MODE_DCT, MODE_DCT, MODE_DCT4X8, MODE_AFV,
                MODE_DCT, MODE_DCT, MODE_DCT, MODE_DCT, MODE_DCT, MODE_DCT]


# ---------------------------------------------------------------- fast_math-inl.h in float32, one rounding per operation
def _eval_rational(x, p, q):
    """rational_polynomial-inl.h:60-97: Horner from the highest coefficient, numerator and denominator, then a division."""
    yp, yq = F(p[-1]), F(q[-1])
    for i in range(len(p) - 2, -1, -1):
        yp = F(F(yp * x) + F(p[i]))
    for i in range(len(q) - 2, -1, -1):
        yq = F(F(yq * x) + F(q[i]))
    return F(yp / yq)


def fast_log2f(x):
    p = [-1.8503833400518310E-06, 1.4287160470083755E+00, 7.4245873327820566E-01]
    q = [9.9032814277590719E-01, 1.0096718572241148E+00, 1.7409343003366853E-01]
    bits = np.array([x], F).view(np.int32)[0]
    exp_bits = np.int32(bits - np.int32(0x3f2aaaab))
    exp_shifted = np.int32(exp_bits >> 23)
    mantissa = np.array([np.int32(bits - np.int32(exp_shifted << 23))], np.int32).view(F)[0]
    return F(_eval_rational(F(mantissa - F(1.0)), p, q) + F(exp_shifted))


def fast_pow2f(x):
    x = F(x)
    floorx = F(np.floor(x))
    exp = np.array([np.int32((np.int32(floorx) + 127) << 23)], np.int32).view(F)[0]
    frac = F(x - floorx)
    num = F(frac + F(1.01749063e+01))
    num = F(F(num * frac) + F(4.88687798e+01))
    num = F(F(num * frac) + F(9.85506591e+01))
    num = F(num * exp)
    den = F(F(frac * F(2.10242958e-01)) + F(-2.22328856e-02))
    den = F(F(den * frac) + F(-1.94414990e+01))
    den = F(F(den * frac) + F(9.85506633e+01))
    return F(num / den)


def fast_powf(base, exponent):
    return fast_pow2f(F(fast_log2f(F(base)) * F(exponent)))


# ---------------------------------------------------------------- quant_weights.cc:94-160
def mult(v):
    v = F(v)
    return F(F(1.0) + v) if v > 0 else F(F(1.0) / F(F(1.0) - v))


def interpolate(pos, mx, array):
    """quant_weights.cc:94-102 (the scalar form the AFV table uses)."""
    n = len(array)
    scaled = F(F(F(pos) * F(n - 1)) / F(mx))
    idx = int(scaled)
    assert idx + 1 < n
    a, b = F(array[idx]), F(array[idx + 1])
    return F(a * fast_powf(F(b / a), F(scaled - F(idx))))


def get_quant_weights(rows, cols, distance_bands, num_bands):
    """quant_weights.cc:129-160: out[c][y][x] for a cols x rows transform from per-channel distance bands."""
    out = np.zeros((3, rows, cols), F)
    for c in range(3):
        bands = [F(distance_bands[c][0])]
        assert bands[0] >= 1e-8
        for i in range(1, num_bands):
            bands.append(F(bands[-1] * mult(distance_bands[c][i])))
            assert bands[-1] >= 1e-8
        scale = F(F(num_bands - 1) / F(F(1.41421356237309504880) + F(1e-6)))
        rcpcol = F(scale / F(cols - 1))
        rcprow = F(scale / F(rows - 1))
        for y in range(rows):
            dy = F(F(y) * rcprow)
            dy2 = F(dy * dy)
            for x in range(cols):
                dx = F(F(x) * rcpcol)
                dist = F(np.sqrt(F(F(dx * dx) + dy2)))  # (MulAdd in the reference: one rounding less, far below the bar)
                if num_bands == 1:
                    w = bands[0]
                else:
                    idx = int(dist)
                    frac = F(dist - F(idx))
                    a, b = bands[idx], bands[idx + 1]
                    w = F(a * fast_powf(F(b / a), frac))
                out[c, y, x] = w
    return out


AFV_FREQS = [None, None, 0.8517778890324296, 5.37778436506804, None, None, 4.734747904497923, 5.449245381693219,
             1.6598270267479331, 4, 7.275749096817861, 10.423227632456525, 2.662932286148962, 7.630657783650829,
             8.962388608184032, 12.97166202570235]


def compute_weights(kind, enc):
    """quant_weights.cc:163-331: the weights (what the reference calls inv_table) of one table, [3][wrows][wcols]; the table
    the decoder multiplies by is 1 / weights."""
    wrows, wcols = 8 * REQ_X[kind], 8 * REQ_Y[kind]
    mode = enc["mode"]
    if mode == MODE_ID:
        assert wrows * wcols == 64
        w = np.zeros((3, 8, 8), F)
        for c in range(3):
            w[c].fill(F(enc["idweights"][c][0]))
            w[c, 0, 1] = w[c, 1, 0] = F(enc["idweights"][c][1])
            w[c, 1, 1] = F(enc["idweights"][c][2])
        return w
    if mode == MODE_DCT2:
        assert wrows * wcols == 64
        w = np.zeros((3, 8, 8), F)
        for c in range(3):
            d = [F(v) for v in enc["dct2weights"][c]]
            w[c, 0, 0] = F(0xBAD)
            w[c, 0, 1] = w[c, 1, 0] = d[0]
            w[c, 1, 1] = d[1]
            w[c, 0:2, 2:4] = d[2]
            w[c, 2:4, 0:2] = d[2]
            w[c, 2:4, 2:4] = d[3]
            w[c, 0:4, 4:8] = d[4]
            w[c, 4:8, 0:4] = d[4]
            w[c, 4:8, 4:8] = d[5]
        return w
    if mode == MODE_DCT4:
        assert wrows * wcols == 64
        w44 = get_quant_weights(4, 4, enc["bands"], enc["num_bands"])
        w = np.zeros((3, 8, 8), F)
        for c in range(3):
            for y in range(8):
                for x in range(8):
                    w[c, y, x] = w44[c, y // 2, x // 2]
            w[c, 0, 1] = F(w[c, 0, 1] / F(enc["dct4multipliers"][c][0]))
            w[c, 1, 0] = F(w[c, 1, 0] / F(enc["dct4multipliers"][c][0]))
            w[c, 1, 1] = F(w[c, 1, 1] / F(enc["dct4multipliers"][c][1]))
        return w
    if mode == MODE_DCT4X8:
        assert wrows * wcols == 64
        w48 = get_quant_weights(4, 8, enc["bands"], enc["num_bands"])
        w = np.zeros((3, 8, 8), F)
        for c in range(3):
            for y in range(8):
                for x in range(8):
                    w[c, y, x] = w48[c, y // 2, x]
            w[c, 1, 0] = F(w[c, 1, 0] / F(enc["dct4x8multipliers"][c]))
        return w
    if mode == MODE_DCT:
        return get_quant_weights(wrows, wcols, enc["bands"], enc["num_bands"])
    if mode == MODE_AFV:
        assert wrows * wcols == 64
        w48 = get_quant_weights(4, 8, enc["bands"], enc["num_bands"])
        w44 = get_quant_weights(4, 4, enc["bands_afv_4x4"], enc["num_bands_afv_4x4"])
        lo = F(0.8517778890324296)
        hi = F(F(F(12.97166202570235) - lo) + F(1e-6))
        w = np.zeros((3, 8, 8), F)
        for c in range(3):
            a = [F(v) for v in enc["afv_weights"][c]]
            bands = [a[5]]
            assert bands[0] >= 1e-8
            for i in range(1, 4):
                bands.append(F(bands[-1] * mult(a[i + 5])))
                assert bands[-1] >= 1e-8
            w[c, 0, 0] = 1
            w[c, 1, 0] = a[0]  # set_weight(0, 1): x = 0, y = 1
            w[c, 0, 1] = a[1]
            w[c, 2, 0] = a[2]
            w[c, 0, 2] = a[3]
            w[c, 2, 2] = a[4]
            for y in range(4):
                for x in range(4):
                    if x < 2 and y < 2:
                        continue
                    w[c, 2 * y, 2 * x] = interpolate(F(F(AFV_FREQS[y * 4 + x]) - lo), hi, bands)
            for y in range(4):  # 4x8 weights in the odd rows, except (1, 0)
                for x in range(8):
                    if x == 0 and y == 0:
                        continue
                    w[c, 2 * y + 1, x] = w48[c, y, x]
            for y in range(4):  # 4x4 weights in even rows / odd columns, except (0, 1)
                for x in range(4):
                    if x == 0 and y == 0:
                        continue
                    w[c, 2 * y, 2 * x + 1] = w44[c, y, x]
        return w
    raise ValueError("mode %d" % mode)


def library_encoding(kind, library):
    """The default table of `kind` from the V(...) lists of the reference's source (tests/golden/ref_constant_floats.json
    'quant_library', extracted by tests/golden/make_float_tables_golden.py in source order)."""
    v = library[kind]
    mode = LIBRARY_MODE[kind]
    e = {"mode": mode}
    if mode == MODE_ID:
        e["idweights"] = [v[3 * c:3 * c + 3] for c in range(3)]
    elif mode == MODE_DCT2:
        e["dct2weights"] = [v[6 * c:6 * c + 6] for c in range(3)]
    elif mode == MODE_DCT4:
        e["num_bands"] = 4
        e["bands"] = [v[4 * c:4 * c + 4] for c in range(3)]
        e["dct4multipliers"] = [v[12 + 2 * c:12 + 2 * c + 2] for c in range(3)]
    elif mode == MODE_DCT4X8:
        e["num_bands"] = 4
        e["bands"] = [v[4 * c:4 * c + 4] for c in range(3)]
        e["dct4x8multipliers"] = v[12:15]
    elif mode == MODE_AFV:
        e["afv_weights"] = [v[9 * c:9 * c + 9] for c in range(3)]
        d48, d44 = library[9], library[3]  # "its bands are DCT4X8's and DCT4X4's" (quant_weights.cc AFV0())
        e["num_bands"] = 4
        e["bands"] = [d48[4 * c:4 * c + 4] for c in range(3)]
        e["num_bands_afv_4x4"] = 4
        e["bands_afv_4x4"] = [d44[4 * c:4 * c + 4] for c in range(3)]
    else:
        nb = len(v) // 3
        e["num_bands"] = nb
        e["bands"] = [v[nb * c:nb * c + nb] for c in range(3)]
    return e


# ---------------------------------------------------------------- the coded form (quant_weights.cc:373-511), written here
class BitWriter:
    """LSB-first bit packer (the order dec_bit_reader.h:84-144 reads in)."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def write(self, nbits, value):
        assert 0 <= value < (1 << nbits)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def f16(self, x):
        """F16Coder (fields.cc:550-575 reads 16 bits: IEEE binary16); returns the value the reader will see."""
        h = np.array([x], np.float16)
        self.write(16, int(h.view(np.uint16)[0]))
        return float(h[0])

    def bytes(self):
        return bytes(self.out + (bytes([self.acc & 0xFF]) if self.n else b""))


def write_dct_params(bw, bands):
    """DecodeDctParams: 4 bits of (count - 1), then the bands per channel as F16; the reader scales the first by 64."""
    nb = len(bands[0])
    bw.write(4, nb - 1)
    seen = [[bw.f16(b) for b in bands[c]] for c in range(3)]
    for c in range(3):
        seen[c][0] = float(F(F(seen[c][0]) * F(64.0)))
    return nb, seen


def write_encoding(bw, kind, spec):
    """One QuantEncoding of the wire format from `spec` (mode + the parameters AS CODED); returns the encoding the decoder
    must arrive at (after F16 rounding and the x64 scalings of quant_weights.cc:373-470)."""
    mode = spec["mode"]
    bw.write(3, mode)
    e = {"mode": mode}
    if mode == MODE_LIBRARY:
        return None  # (kNumPredefinedTables = 1: no selector bits) -> the caller substitutes the library table
    if mode == MODE_ID:
        e["idweights"] = [[float(F(F(bw.f16(v)) * F(64))) for v in spec["idweights"][c]] for c in range(3)]
    elif mode == MODE_DCT2:
        e["dct2weights"] = [[float(F(F(bw.f16(v)) * F(64))) for v in spec["dct2weights"][c]] for c in range(3)]
    elif mode == MODE_DCT4X8:
        e["dct4x8multipliers"] = [bw.f16(spec["dct4x8multipliers"][c]) for c in range(3)]
        e["num_bands"], e["bands"] = write_dct_params(bw, spec["bands"])
    elif mode == MODE_DCT4:
        e["dct4multipliers"] = [[bw.f16(v) for v in spec["dct4multipliers"][c]] for c in range(3)]
        e["num_bands"], e["bands"] = write_dct_params(bw, spec["bands"])
    elif mode == MODE_AFV:
        e["afv_weights"] = []
        for c in range(3):
            a = [bw.f16(v) for v in spec["afv_weights"][c]]
            e["afv_weights"].append([float(F(F(a[i]) * F(64))) if i < 6 else a[i] for i in range(9)])
        e["num_bands"], e["bands"] = write_dct_params(bw, spec["bands"])
        e["num_bands_afv_4x4"], e["bands_afv_4x4"] = write_dct_params(bw, spec["bands_afv_4x4"])
    elif mode == MODE_DCT:
        e["num_bands"], e["bands"] = write_dct_params(bw, spec["bands"])
    else:
        raise ValueError("mode %d" % mode)
    return e


# ---------------------------------------------------------------- ac_strategy.cc:28-79
def natural_order(covered_x, covered_y):
    """ComputeNaturalCoeffOrder of a strategy covering covered_x x covered_y blocks: order[k] = position (in the cx * 8 wide
    coefficient layout) of the k-th coefficient of the scan."""
    cx, cy = covered_x, covered_y
    if cy > cx:  # CoefficientLayout (coeff_order_fwd.h): the longer side becomes the row length
        cx, cy = cy, cx
    xs = cx // cy
    xsm = xs - 1
    xss = (xs - 1).bit_length()  # CeilLog2Nonzero of a power of two
    out = [None] * (cx * cy * 64)
    cur = cx * cy
    n = cx * 8
    for i in range(n):
        for j in range(i + 1):
            x, y = j, i - j
            if i % 2:
                x, y = y, x
            if y & xsm:
                continue
            y >>= xss
            if x < cx and y < cy:
                val = y * cx + x
            else:
                val = cur
                cur += 1
            out[val] = y * cx * 8 + x
    for ip in range(n - 1, 0, -1):
        i = ip - 1
        for j in range(i + 1):
            x = n - 1 - (i - j)
            y = n - 1 - j
            if i % 2:
                x, y = y, x
            if y & xsm:
                continue
            y >>= xss
            out[cur] = y * cx * 8 + x
            cur += 1
    assert cur == len(out) and None not in out
    return out


def zero_density_context(nonzeros_left, k, log2_covered, prev, freq_ctx, nnz_ctx):
    """ac_context.h:63-84 with kCoeffFreqContext / kCoeffNumNonzeroContext as the reference's source lists them."""
    covered = 1 << log2_covered
    nonzeros_left = (nonzeros_left + covered - 1) >> log2_covered
    k >>= log2_covered
    return (nnz_ctx[nonzeros_left & 63] + freq_ctx[k & 63]) * 2 + prev
