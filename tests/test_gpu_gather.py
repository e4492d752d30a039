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
    info = g.info()                                    # what the communicator says about itself
    assert info["ranks"] == 1 and info["version"] >= 20000, info
    g.close()
    for p_ in (iq_d, pcm_d, recv_d):
        eng.dev_free(p_)


def test_gather_of_two_ranks_in_one_process():
    """The multi-rank branch of iqd_gather_pcm - a grouped ncclRecv per peer on the root, ncclSend elsewhere - with two
    engines on two GPUs, one host thread each (communicator set-up is collective), the root being rank 1.  Needs two
    visible devices: skipped on the one-GPU boxes this builder has."""
    import threading
    from rtlsdrdiags_amd import capi
    if capi._lib().iqd_device_count() < 2:
        pytest.skip("needs two GPUs")
    n_ch, n, world, root = 4, 1 << 15, 2, 1
    uid = capi.Gatherer.unique_id()
    nb = n_ch * (n // 32) * 2
    out, errors = {}, []

    def rank_thread(rank):
        try:
            eng = capi.Engine(n_ch, device=rank)
            eng.set_mode("fm")
            u8 = np.stack([synth.fm_tone(n, seed=70 + 10 * rank + c) for c in range(n_ch)])
            iq_d, pcm_d = eng.dev_alloc(u8.nbytes), eng.dev_alloc(nb)
            recv_d = eng.dev_alloc(world * (nb + 64)) if rank == root else 0
            eng.dev_upload(iq_d, u8)
            g = capi.Gatherer(eng, uid, rank, world, root)
            assert g.info()["ranks"] == world
            for _ in range(2):                                   # two steps: the second is ordered behind the first gather
                eng.accept_device(iq_d, 2 * n, pcm_d)
                g.gather(pcm_d, [nb] * world, recv_d, nb + 64)
            eng.synchronize()
            out[("direct", rank)] = eng.dev_download(pcm_d, nb, np.int16)
            if rank == root:
                out["rows"] = eng.dev_download(recv_d, world * (nb + 64), np.uint8).reshape(world, nb + 64)[:, :nb].copy()
            g.close()
            eng.close()
        except Exception as exc:                                 # (a failing thread must not leave the other one waiting forever)
            errors.append((rank, repr(exc)))

    threads = [threading.Thread(target=rank_thread, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in threads), errors
    for r in range(world):
        assert np.array_equal(out["rows"][r].view(np.int16), out[("direct", r)]), r
