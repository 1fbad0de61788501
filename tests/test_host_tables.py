"""The host-only rows of BOTH front-ends against independent NumPy readings written from the
reference's text (tests/host_tables_np.py): dequantisation tables from the default library and from coded DequantMatrices
sections of every mode the format has (RAW is refused by both front-ends), natural coefficient orders of the 13
order buckets, the zero-density context arithmetic. The product's host front-end and the oracle are near twins for these
rows, so neither is the other's check; this third reading shares no code with them."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import host_tables_np as T
from test_kats import ROOT, _binary


def _read_items(path):
    raw = open(path, "rb").read()
    items, pos = [], 0
    while pos < len(raw):
        (n,) = struct.unpack_from("<I", raw, pos)
        items.append(raw[pos + 4:pos + 4 + 4 * n])
        pos += 4 + 4 * n
    return items


def _dump(which, tmp_path, stream=None):
    out = os.path.join(str(tmp_path), "tables_%s.bin" % which)
    cmd = [_binary(which), "tables", out]
    if stream is not None:
        sp = os.path.join(str(tmp_path), "stream.bin")
        open(sp, "wb").write(stream)
        cmd.append(sp)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    items = _read_items(out)
    assert len(items) == 17 + 13 + 1
    tables = [np.frombuffer(b, np.float32) for b in items[:17]]
    orders = [np.frombuffer(b, np.uint32) for b in items[17:30]]
    ctx = np.frombuffer(items[30], np.uint32).reshape(-1, 5)
    return tables, orders, ctx


def _check_tables(tables, encodings):
    for kind in range(17):
        want = (np.float32(1.0) / T.compute_weights(kind, encodings[kind])).reshape(-1)
        got = tables[kind]
        assert got.shape == want.shape, kind
        rel = np.abs(got.astype(np.float64) - want) / np.abs(want)
        assert rel.max() <= 1e-6, "table %d: entry %d is %.9g, the reference's text gives %.9g" % (
            kind, int(rel.argmax()), got[rel.argmax()], want[rel.argmax()])


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_default_tables_orders_and_contexts_match_the_reading_of_the_reference_text(which, tmp_path):
    """All 17 default dequantisation tables (x 3 channels), built by the NumPy generator from the library parameters AS THE
    REFERENCE'S SOURCE LISTS THEM (ref_constant_floats.json), within 1e-6 relative (float32 rounding of the same rational
    approximations); the 13 natural orders exactly; the zero-density context of ~60 000 (covered, k, non-zeros, prev)
    samples exactly. This exercises what the uniform-table vector never does: band interpolation with several bands, the
    Identity / DCT2 / DCT4 / DCT4x8 / AFV layouts, every rectangular zig-zag."""
    floats = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constant_floats.json")))
    consts = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constant_tables.json")))
    tables, orders, ctx = _dump(which, tmp_path)
    _check_tables(tables, [T.library_encoding(k, floats["quant_library"]) for k in range(17)])
    for ord_ in range(13):
        s = consts["kStrategyOrder"].index(ord_)
        want = T.natural_order(consts["covered_blocks_x"][s], consts["covered_blocks_y"][s])
        assert orders[ord_].tolist() == want, "order bucket %d" % ord_
    assert len(ctx) > 20000
    for log2c, k, nz, prev, got in ctx[::7].tolist():
        assert got == T.zero_density_context(nz, k, log2c, prev, consts["kCoeffFreqContext"], consts["kCoeffNumNonzeroContext"])


def _random_spec(rng, kind):
    def bands(nb):
        return [[float(rng.uniform(0.3, 60.0))] + [float(rng.uniform(-0.6, 1.5)) for _ in range(nb - 1)] for _ in range(3)]

    single = T.REQ_X[kind] * T.REQ_Y[kind] == 1
    mode = int(rng.choice([T.MODE_ID, T.MODE_DCT2, T.MODE_DCT4, T.MODE_DCT4X8, T.MODE_AFV, T.MODE_DCT, T.MODE_LIBRARY])) if single else \
        int(rng.choice([T.MODE_DCT, T.MODE_DCT, T.MODE_LIBRARY]))
    spec = {"mode": mode}
    if mode == T.MODE_ID:
        spec["idweights"] = [[float(rng.uniform(0.5, 40)) for _ in range(3)] for _ in range(3)]
    elif mode == T.MODE_DCT2:
        spec["dct2weights"] = [[float(rng.uniform(0.5, 40)) for _ in range(6)] for _ in range(3)]
    elif mode == T.MODE_DCT4:
        spec["dct4multipliers"] = [[float(rng.uniform(0.5, 2.0)) for _ in range(2)] for _ in range(3)]
        spec["bands"] = bands(int(rng.integers(1, 6)))
    elif mode == T.MODE_DCT4X8:
        spec["dct4x8multipliers"] = [float(rng.uniform(0.5, 2.0)) for _ in range(3)]
        spec["bands"] = bands(int(rng.integers(1, 6)))
    elif mode == T.MODE_AFV:
        spec["afv_weights"] = [[float(rng.uniform(0.5, 40)) for _ in range(6)] + [float(rng.uniform(-0.5, 1.0)) for _ in range(3)] for _ in range(3)]
        spec["bands"] = bands(int(rng.integers(1, 6)))
        spec["bands_afv_4x4"] = bands(int(rng.integers(1, 6)))
    elif mode == T.MODE_DCT:
        spec["bands"] = bands(int(rng.integers(1, 17)))  # (the count is coded in 4 bits: 1..16)
    return spec


@pytest.mark.parametrize("which", ["product", "oracle"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_coded_tables_of_every_mode_match_the_reading_of_the_reference_text(which, seed, tmp_path):
    """A coded DequantMatrices section (quant_weights.cc:497-511) with every table in a randomly chosen mode and randomly
    drawn parameters -- Identity, DCT2, DCT4, DCT4x8, AFV and distance bands with 1..16 entries -- written by this test's own
    bit writer, read by the front-end's reader and turned into tables by its generator: equal, within 1e-6 relative, to the
    NumPy generator run on the parameters as the wire format carries them (binary16, seeds x 64)."""
    floats = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constant_floats.json")))
    rng = np.random.default_rng(seed)
    bw = T.BitWriter()
    bw.write(1, 0)  # not all default
    encodings = []
    for kind in range(17):
        e = T.write_encoding(bw, kind, _random_spec(rng, kind))
        encodings.append(e if e is not None else T.library_encoding(kind, floats["quant_library"]))
    tables, _, _ = _dump(which, tmp_path, stream=bw.bytes())
    _check_tables(tables, encodings)
