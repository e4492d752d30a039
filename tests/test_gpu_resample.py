"""GPU (MI355X): the stand-alone resamplers (float Decimator / Interpolator, Interpolator_int16)
against the reference's golden vectors and the oracle: bit-identical floats, uneven call boundaries (filter state
and decimator phase carried), several channels at once."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KIND = {"dec": "decimate_f32", "int": "interpolate_f32", "i16": "interpolate_i16"}


@pytest.fixture(scope="module")
def capi():
    from rtlsdrdiags_amd import capi as c
    return c


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
@pytest.mark.parametrize("what", ["dec", "int", "i16", "i16sat"])
def test_golden_streams_in_uneven_calls(capi, golden, name, what):
    g = golden["resample"]
    h, f = g["h_" + name], int(g["f_" + name])
    x = g["x"] if what in ("dec", "int") else (g["x16"] if what == "i16" else g["xsat"])
    if what == "i16sat":
        h = np.clip(4 * h, -1, 0.99997).astype(np.float32)
    eng = capi.Engine(1)
    r = capi.Resampler(eng, KIND[what[:3]], h, f)
    cuts = [0, 1, 2, 5, 700, 701, 1999, len(x)]
    out = np.concatenate([r.run(x[a:b])[0] for a, b in zip(cuts[:-1], cuts[1:])])
    want = g[what + "_" + name]
    assert out.dtype == want.dtype and np.array_equal(out.view(np.uint8), want.view(np.uint8))
    r.reset()                                   # resetFilterState(): the stream starts over
    again = r.run(x)[0]
    assert np.array_equal(again.view(np.uint8), want.view(np.uint8))
    r.close()


def test_many_channels_against_the_oracle(capi, oracle):
    rng = np.random.default_rng(3)
    n_ch, n = 300, 4096
    x = rng.normal(0, 500, (n_ch, n)).astype(np.float32)
    h = rng.normal(0, 0.2, 48).astype(np.float32)
    eng = capi.Engine(1)
    dec, itp = capi.Resampler(eng, "decimate_f32", h, 5, n_ch), capi.Resampler(eng, "interpolate_f32", h, 6, n_ch)
    d = np.concatenate([dec.run(x[:, :1001]), dec.run(x[:, 1001:])], axis=1)
    u = np.concatenate([itp.run(x[:, :333]), itp.run(x[:, 333:])], axis=1)
    for c in range(0, n_ch, 37):
        assert np.array_equal(d[c].view(np.uint32), oracle.decimate_f32(h, 5, x[c]).view(np.uint32)), c
        assert np.array_equal(u[c].view(np.uint32), oracle.interpolate_f32(h, 6, x[c]).view(np.uint32)), c
