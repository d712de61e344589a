"""relate_distance_matrix_noalloc (the entry point for callers that cannot allocate) against
relate_plan_distance_matrix, through the C ABI: same kernel, so every float must be the same; two calls in a
row, the second on a smaller panel (the argument array keeps the first call's contents), and the size limit."""
import numpy as np
import pytest

import synth_np as sy
from test_gpu_distance import bookkeeping

pytestmark = pytest.mark.gpu


def _topology(ra, torch, N, L, W, seed, theta=0.001):
    seq, r, wb = sy.make_chunk(N=N, L=L, W=W, seed=seed, r_mode="flat")
    rpos = np.concatenate([[0.0], np.cumsum(r / 2500.0)])
    plan = ra.Plan(0, 1, 0)
    plan.load_host(seq, r, wb)
    plan.prepare(theta, theta, 0, N)
    stones = torch.zeros(plan.out_bytes(), dtype=torch.uint8, device="cuda:0")
    plan.paint(stones.data_ptr())
    w = W - 1
    rows, pitch, off = plan.repaint_layout(w)
    top = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda:0")
    ls = torch.zeros((rows,), dtype=torch.float32, device="cuda:0")
    R = ra.record_bytes(N)
    plan.repaint(w, stones.data_ptr() + w * N * R, top.data_ptr(), ls.data_ptr())
    import oracle_lib as ol
    a_snp = ol.decode_records(stones.cpu().numpy().reshape(plan.W, N, R)[w:w + 1])["a_snp"][0]
    snp = (int(wb[w]) + int(wb[w + 1]) - 1) // 2
    row, interp, wl, wr = bookkeeping(seq, rpos, wb, w, snp, a_snp, off)
    return plan, top, ls, rows, pitch, row, interp, wl, wr


def test_noalloc_entry_point_matches_the_plan_entry_point():
    import torch
    import relate_amd as ra
    from relate_amd.paint import distance_matrix_noalloc
    for N, L, seed in ((700, 500, 81), (130, 400, 82)):  # the second call is the smaller one
        plan, top, ls, rows, pitch, row, interp, wl, wr = _topology(ra, torch, N, L, 2, seed)
        assert interp.any() and not interp.all()
        want = torch.zeros((N, N), dtype=torch.float32, device="cuda:0")
        got = torch.full((N, N), -1.0, dtype=torch.float32, device="cuda:0")
        plan.distance_matrix(N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, interp, wl, wr, want.data_ptr())
        # the first call on a device looks up the address of the library's argument array (and makes the runtime
        # load that code object): set-up, once; every call after it must leave the device's free memory alone
        distance_matrix_noalloc(0, N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, interp, wl, wr, got.data_ptr())
        torch.cuda.synchronize()
        got.fill_(-1.0)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        distance_matrix_noalloc(0, N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, interp, wl, wr, got.data_ptr())
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the call must not allocate device memory"
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        # rows taken as they are need no weights
        none = np.zeros(N, dtype=np.uint8)
        plan.distance_matrix(N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, none, wl, wr, want.data_ptr())
        distance_matrix_noalloc(0, N, top.data_ptr(), pitch, ls.data_ptr(), rows, row, none, None, None, got.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        plan.close()


def test_noalloc_entry_point_refuses_what_does_not_fit():
    import torch
    import relate_amd as ra
    from relate_amd.paint import distance_matrix_noalloc
    N = 36865
    dummy = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    row = np.zeros(N, dtype=np.int64)
    flag = np.zeros(N, dtype=np.uint8)
    with pytest.raises(ra.RelateError) as e:
        distance_matrix_noalloc(0, N, dummy.data_ptr(), N, dummy.data_ptr(), 1, row, flag, None, None, dummy.data_ptr())
    assert e.value.code == -5 and "36864" in str(e.value)
    with pytest.raises(ra.RelateError) as e:  # a row index past the topology
        distance_matrix_noalloc(0, 4, dummy.data_ptr(), 4, dummy.data_ptr(), 1, np.array([0, 0, 1, 0], dtype=np.int64),
                                flag[:4], None, None, dummy.data_ptr())
    assert e.value.code == -1
