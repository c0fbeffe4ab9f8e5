"""CPU, world_size 2, gloo: the utterance-sharding + all-gather path (dex_tts_amd/dist.py).  The sampler
itself is injected (a deterministic stand-in here — the HIP engine needs a GPU); what is tested is the
partition, the fixed global padding, the gather order, and equality with the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dex_tts_amd import dist as D, synth


def fake_sampler(z, mask, mu):
    # per-utterance, padding-length dependent (like the real net): mean over ALL columns enters the result
    return z * 0.5 + mu.mean(dim=(1, 2), keepdim=True) + mask * 0.25


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def fake_dex_sampler(z, mask, mu, ref=None, sty=None, sty_lengths=None):
    # the DEX inputs enter per utterance, like the style adaptors: a wrong row in any of them changes that utterance's result
    s = sum(r.mean(dim=(1, 2)) * (j + 1) for j, r in enumerate(ref)) + sty.amax(dim=(1, 2)) + sty_lengths.to(torch.float32) * 0.01
    return fake_sampler(z, mask, mu) + s[:, None, None]


def _dex_extras(B):
    ref, _, sty, sl = synth.make_dex_style(B, 12, 16, 8, sty_lengths=[16 - (i % 5) for i in range(B)])
    return {"ref": [torch.from_numpy(r) for r in ref], "sty": torch.from_numpy(sty), "sty_lengths": torch.from_numpy(sl)}


def _worker(rank, world, port, lengths, T, q, local=False, dex=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mu, mask, z, _ = synth.make_inputs(len(lengths), T, lengths)
    mu, mask, z = map(torch.from_numpy, (mu, mask, z))
    extras = _dex_extras(len(lengths)) if dex else None
    if local:       # a rank holds only its own utterances (in shard order) plus the lengths of all of them
        mu, mask, z = (D.take_shard(t, lengths) for t in (mu, mask, z))
        assert mu.shape[0] == len(D.partition(lengths, world)[rank])
        if dex:
            extras = {k: D.take_shard(v, lengths) for k, v in extras.items()}
    full = D.sample_sharded(fake_dex_sampler if dex else fake_sampler, mu, mask, z, lengths, local=local, extras=extras)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_balanced():
    lengths = [100, 20, 300, 40, 250, 60, 10]
    sh = D.partition(lengths, 2)
    assert sorted(sum(sh, [])) == list(range(7))
    loads = [sum(lengths[i] for i in s) for s in sh]
    assert abs(loads[0] - loads[1]) <= max(lengths)
    assert D.padded_length(lengths) == 300 and D.padded_length([301]) == 304


@pytest.mark.parametrize("lengths,local,dex", [([64, 40, 52, 30, 64], False, False), ([16], False, False), ([64, 40, 52, 30, 64], True, False),
                                               ([48, 48, 20, 36], True, False),
                                               ([64, 40, 52, 30, 64], False, True), ([20, 48, 36, 48, 8], True, True)])   # DEX style inputs ride `extras`
def test_sharded_equals_unsharded(lengths, local, dex):
    T = D.padded_length(lengths)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, T, q, local, dex)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mu, mask, z, _ = synth.make_inputs(len(lengths), T, lengths)
    if dex:
        ref = fake_dex_sampler(*map(torch.from_numpy, (z, mask, mu)), **_dex_extras(len(lengths))).numpy()
    else:
        ref = fake_sampler(*map(torch.from_numpy, (z, mask, mu))).numpy()
    for r in range(world):
        np.testing.assert_array_equal(got[r], ref)


@pytest.mark.parametrize("local", [False, True])
def test_single_process_keeps_input_order(local):
    """world == 1 (no process group): with ``local=True`` the caller hands the rows in SHARD order (length-sorted, what
    ``take_shard`` returns) and the result must still come back in INPUT order (it came back sorted)."""
    lengths = [20, 64, 36, 64, 8, 52]                     # deliberately not sorted
    T = D.padded_length(lengths)
    mu, mask, z, _ = synth.make_inputs(len(lengths), T, lengths)
    mu, mask, z = map(torch.from_numpy, (mu, mask, z))
    ref = fake_sampler(z, mask, mu).numpy()
    if local:
        order = D.partition(lengths, 1)[0]
        assert order != sorted(order)
        mu, mask, z = (D.take_shard(t, lengths) for t in (mu, mask, z))
    got = D.sample_sharded(fake_sampler, mu, mask, z, lengths, local=local).numpy()
    np.testing.assert_array_equal(got, ref)


def test_length_buckets_are_a_partition_with_their_own_padding():
    lengths = [100, 20, 300, 40, 250, 60, 10, 299, 64, 65]
    b = D.buckets_of(lengths, 64)
    assert sorted(i for _, idx in b for i in idx) == list(range(len(lengths)))
    for Tb, idx in b:
        assert Tb % 4 == 0 and Tb >= max(lengths[i] for i in idx) and Tb - max(lengths[i] for i in idx) < 4
        assert len({-(-lengths[i] // 64) for i in idx}) == 1                 # one rounded-up length per bucket
    assert [Tb for Tb, _ in b] == sorted(Tb for Tb, _ in b)
    with pytest.raises(ValueError):
        D.buckets_of(lengths, 0)


def test_bucketed_sampling_equals_the_sampler_on_each_bucket():
    """Opt-in length bucketing (single process): per bucket the result is the sampler run on that bucket alone at the bucket's own
    padded length - and it differs from the globally padded run (the stand-in, like the real net, depends on the padded columns)."""
    lengths = [120, 33, 128, 70, 64, 90]
    T = D.padded_length(lengths)
    mu, mask, z, _ = synth.make_inputs(len(lengths), T, lengths)
    mu, mask, z = map(torch.from_numpy, (mu, mask, z))
    got = D.sample_bucketed(fake_sampler, mu, mask, z, lengths, bucket_width=64)
    glob = fake_sampler(z, mask, mu)
    seen = 0
    for Tb, idx in D.buckets_of(lengths, 64):
        ix = torch.tensor(idx)
        want = fake_sampler(z[ix][:, :, :Tb], mask[ix][:, :, :Tb], mu[ix][:, :, :Tb])
        assert torch.equal(got[ix][:, :, :Tb], want)
        assert float(got[ix][:, :, Tb:].abs().max() if Tb < T else 0.0) == 0.0
        if Tb < T:
            assert not torch.equal(want, glob[ix][:, :, :Tb])
            seen += 1
    assert seen >= 1
