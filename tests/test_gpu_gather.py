"""GPU (MI355X): the engine's own PCM gather (iqd_gather_*, RCCL on the engine's stream) in the one shape a single-GPU
box can run - a communicator of one rank - and the API's argument checks.  The multi-rank transfers are the same calls
with ncclSend / ncclRecv inside one group; bench.py --gather runs them on the driver's multi-GPU node."""
import numpy as np
import pytest

from rtlsdrdiags_amd import synth

pytestmark = pytest.mark.gpu


def test_gather_of_one_rank_delivers_the_pcm_behind_the_accept():
    from rtlsdrdiags_amd import capi
    n_ch, n = 8, 1 << 15
    u8 = np.stack([synth.fm_tone(n, seed=40 + c) for c in range(n_ch)])
    eng = capi.Engine(n_ch)
    eng.set_mode("fm")
    iq_d, pcm_d, recv_d = eng.dev_alloc(u8.nbytes), eng.dev_alloc(n_ch * (n // 32) * 2), eng.dev_alloc(n_ch * (n // 32) * 2 + 64)
    eng.dev_upload(iq_d, u8)
    g = capi.Gatherer(eng, capi.Gatherer.unique_id(), 0, 1, 0)
    nb = n_ch * (n // 32) * 2
    eng.accept_device(iq_d, 2 * n, pcm_d)
    g.gather(pcm_d, [nb], recv_d, nb + 64)            # queued behind the accept, no synchronisation in between
    eng.synchronize()
    got = eng.dev_download(recv_d, nb, np.int16).reshape(n_ch, -1)
    direct = eng.dev_download(pcm_d, nb, np.int16).reshape(n_ch, -1)
    assert np.array_equal(got, direct) and got.any()
    with pytest.raises(capi.IqdError):
        g.gather(pcm_d, [nb], recv_d, nb - 2)         # a row that does not fit its stride
    g.close()
    for p_ in (iq_d, pcm_d, recv_d):
        eng.dev_free(p_)
