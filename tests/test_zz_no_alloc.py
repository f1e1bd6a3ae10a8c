"""The compaction entry point that allocates nothing (gtsc_compact_device_noalloc): its results must
equal gtsc_compact_device's, on two calls in a row, the second with a smaller input (the static
scratch array keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import pygtsc as G  # noqa: E402
from pygtsc import synth  # noqa: E402


def compact_both_ways(ctx, b):
    S = b.nseries
    lens = np.diff(b.pt_off.astype(np.int64))
    slot = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum([G.max_words(int(p)) for p in lens], out=slot[1:])
    up = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda()
    d_block, d_ts, d_val, d_po, d_so = map(up, (b.block_ts, b.ts, b.val, b.pt_off, slot))
    d_out = torch.zeros(int(slot[-1]), dtype=torch.int64, device="cuda")
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, pt_off=d_po, out_off=d_so)
    ctx.sync()
    total = int(d_nw.sum())
    res = []
    for noalloc in (False, True):
        d_c = torch.full((total,), -1, dtype=torch.int64, device="cuda")
        d_co = torch.full((S + 1,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.compact_device(d_out, d_nw, d_c, d_co, slot_off=d_so, noalloc=noalloc)
        ctx.sync()
        res.append((d_c.cpu().numpy(), d_co.cpu().numpy()))
    assert int(res[0][1][-1]) == total
    assert np.array_equal(res[0][1], res[1][1]), "offsets differ"
    assert np.array_equal(res[0][0], res[1][0]), "compacted words differ"
    return total


def test_noalloc_compaction_equals_the_allocating_one():
    ctx = G.Context(0)
    try:
        # many scan tiles first (partial sums fill a stretch of the scratch array), then a small batch
        big = compact_both_ways(ctx, synth.ragged_mix(seed=3, nseries=70000, pmax=12))
        small = compact_both_ways(ctx, synth.ragged_mix(seed=4, nseries=900, pmax=40))
        assert big > small > 0
        # the point of the entry point: it allocates nothing.  The device's free memory is the same before and
        # after calls on a warm context, for a batch larger than any before (the allocating form grows its
        # scratch buffer for such a batch; torch's own allocations are made beforehand).
        S = 300000
        d_nw = torch.full((S,), 3, dtype=torch.int64, device="cuda")
        d_slots = torch.zeros(S * 3, dtype=torch.int64, device="cuda")
        d_c = torch.zeros(S * 3, dtype=torch.int64, device="cuda")
        d_co = torch.zeros(S + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            ctx.compact_device(d_slots, d_nw, d_c, d_co, uniform_slot_words=3, noalloc=True)
        ctx.sync()
        assert torch.cuda.mem_get_info()[0] == free_before, "gtsc_compact_device_noalloc changed the device's free memory"
        assert int(d_co[-1]) == 3 * S
        # the limit is enforced before anything is launched
        with pytest.raises(G.GtscError) as ei:
            z = torch.zeros(8, dtype=torch.int64, device="cuda")
            rc = G.lib().gtsc_compact_device_noalloc(ctx._h, G.NOALLOC_MAX_SERIES + 1, z.data_ptr(), None, 3, z.data_ptr(),
                                                     z.data_ptr(), z.data_ptr())
            G._check(rc)
        assert ei.value.code == G.ERR_ARG
    finally:
        ctx.close()
