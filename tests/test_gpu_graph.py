"""Replay of the training / inference tower (trainval.use_graph) in both replay modes -- the recorded launch plan (csrc/plan.cc:
"plan") and the captured HIP graph (True): same results as eager launches, a fresh dropout mask on every replay, micro-step
accumulation across replays, bounded caches; and the launch plan with the second stream and an RCCL call inside it."""
import numpy as np
import pytest
import torch

import dgcnn
from dgcnn import _engine as E

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["plan", True], ids=["launch-plan", "hip-graph"])
def mode(request):
    return request.param


def _flags(**kw):
    base = dict(EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[64, 64], KVALUE=10, FC_LAYERS=2, FC_FILTERS=[128, 64], NUM_CLASS=2,
                NUM_CHANNEL=3, TRAIN=True, SEED=11, LEARNING_RATE=1e-3)
    base.update(kw)
    return dgcnn.DGCNN_FLAGS(**base)


def _data(steps=4, B=4, N=512, seed=0):
    rng = np.random.default_rng(seed)
    return (torch.from_numpy(rng.random((steps, B, N, 3), dtype=np.float32)).cuda(),
            torch.from_numpy(rng.integers(0, 2, (steps, B, N)).astype(np.int32)).cuda())


def _train(graph, steps, pts, lab, **kw):
    tv = dgcnn.trainval(_flags(**kw)).initialize().use_graph(graph)
    losses = []
    for s in range(steps):
        tv.zero_gradients(None)
        for micro in range(2):                                   # two accumulated micro-steps per update
            res = tv.accum_gradient(None, [pts[s]], [lab[s]])
            losses.append(float(res[2]))
        tv.apply_gradient(None)
    return tv, np.asarray(losses), dgcnn.ctx().flat_param.cpu().numpy().copy()


def test_graph_replay_matches_eager_training(mode):
    """Same dropout stream (the seed advances per tower call in both modes), same arithmetic.  The default kernels are not
    bit-reproducible run to run (atomically accumulated BatchNorm sums -> last-bit feature differences -> a few different
    layer-1 neighbour lists -> Adam amplifies; two EAGER runs differ by the same amount, which made a noise-relative bar
    flaky): the comparison runs the deterministic kernels, where graph replay must retrace the eager run exactly."""
    pts, lab = _data()
    tv_e, loss_e, p_e = _train(False, 4, pts, lab, DETERMINISTIC=True)
    tv_g, loss_g, p_g = _train(mode, 4, pts, lab, DETERMINISTIC=True)
    assert len(tv_g._graphs) == 1 and not tv_e._graphs            # one (shape, mode) -> one captured graph, replayed 6 times
    np.testing.assert_allclose(loss_g, loss_e, rtol=0, atol=1e-6)             # (a wrong mask stream moves the loss by ~1e-2)
    d = np.abs(p_g - p_e)
    print("graph vs eager (deterministic kernels): max |dp| %.2e" % d.max())
    assert d.max() <= 1e-6, d.max()
    assert np.abs(p_g - dgcnn.trainval(_flags()).initialize()._ctx.flat_param.cpu().numpy()).max() > 1e-3   # it trained
    # default kernels: the first two micro-steps (before any update) still agree to the atomics' noise
    tv_e, loss_e, _ = _train(False, 1, pts, lab, DETERMINISTIC=False)
    tv_g, loss_g, _ = _train(mode, 1, pts, lab, DETERMINISTIC=False)
    np.testing.assert_allclose(loss_g, loss_e, rtol=0, atol=2e-5)


def test_graph_replay_draws_a_new_dropout_mask_and_accumulates(mode):
    pts, lab = _data(steps=1)
    tv = dgcnn.trainval(_flags()).initialize().use_graph(mode)
    c = dgcnn.ctx()
    losses, grads = [], []
    for i in range(4):                                           # call 0 eager, call 1 captures + replays, 2..3 replay
        tv.zero_gradients(None)
        res = tv.accum_gradient(None, [pts[0]], [lab[0]])
        losses.append(float(res[2]))
        grads.append(c.flat_grad.cpu().numpy().copy())
    assert len(tv._graphs) == 1
    assert len({round(l, 7) for l in losses}) == 4, losses        # same inputs, same weights: only the mask differs
    # without dropout the replays are repeatable and two replays without zeroing accumulate 2x (deterministic kernels: with
    # the atomically summed statistics two replays differ by a few flipped near-tie neighbours, ~2e-3 of the gradient norm)
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    try:
        tv = dgcnn.trainval(_flags(DETERMINISTIC=True)).initialize().use_graph(mode)
        c = dgcnn.ctx()
        for i in range(2):
            tv.zero_gradients(None)
            tv.accum_gradient(None, [pts[0]], [lab[0]])
        g1 = c.flat_grad.cpu().numpy().copy()
        tv.accum_gradient(None, [pts[0]], [lab[0]])
        g2 = c.flat_grad.cpu().numpy().copy()
        assert np.linalg.norm(g2 - 2 * g1) <= 1e-6 * np.linalg.norm(g1)
    finally:
        E.DROPOUT_KEEP = keep
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT


def test_graph_replay_inference_and_new_shapes(mode):
    pts, lab = _data(steps=3)
    f = _flags(TRAIN=False)
    outs = []
    tv = dgcnn.trainval(f).initialize().use_graph(mode)
    for s in range(3):
        sm = tv.inference(None, [pts[s]], [lab[s]])
        outs.append([o.clone() for o in sm])
    assert len(tv._graphs) == 1
    tv.use_graph(False)
    for s in range(3):
        want = tv.inference(None, [pts[s]], [lab[s]])
        np.testing.assert_allclose(outs[s][0].cpu().numpy(), want[0].cpu().numpy(), rtol=0, atol=2e-4)
        assert abs(float(outs[s][-1]) - float(want[-1])) < 2e-4
    tv.use_graph(mode)
    other = torch.rand(2, 300, 3, device="cuda")                 # a shape seen once runs eagerly, no graph is kept for it
    tv.inference(None, [other])
    assert len(tv._graphs) == 1


def test_graph_outputs_of_several_towers_and_micro_steps_do_not_alias(mode):
    """Every replay of one (shape, mode) key writes the same static buffers: the softmax / [loss, accuracy] handed to the
    caller must be copies, or every tower of a multi-tower call (and every gathered micro-step) shows the LAST replay."""
    pts, lab = _data(steps=4, B=2, N=384, seed=3)
    towers_p, towers_l = [pts[0], pts[1]], [lab[0], lab[1]]
    f = _flags(TRAIN=False)
    tv = dgcnn.trainval(f).initialize().use_graph(False)
    want = tv.inference(None, towers_p, towers_l)
    tv.use_graph(mode)
    got = None
    for _ in range(3):                                            # eager sighting, capture + replay, replay
        got = tv.inference(None, towers_p, towers_l)
    assert len(tv._graphs) == 1
    assert float((got[0] - got[1]).abs().max()) > 1e-3           # two different towers
    for t in range(2):
        np.testing.assert_allclose(got[t].cpu().numpy(), want[t].cpu().numpy(), rtol=0, atol=2e-4)
    assert abs(float(got[-1]) - float(want[-1])) < 2e-4 and abs(float(got[-2]) - float(want[-2])) < 1e-6   # tower MEAN loss / accuracy

    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    try:
        def micro_losses(graph):
            tv = dgcnn.trainval(_flags()).initialize().use_graph(graph)
            tv.zero_gradients(None)
            tv.accum_gradient(None, [pts[0]], [lab[0]])           # (graph mode: the eager sighting)
            tv.zero_gradients(None)
            res = [tv.accum_gradient(None, [pts[s]], [lab[s]]) for s in range(4)]      # results read AFTER all micro-steps
            tv.apply_gradient(None)
            return [float(r[2]) for r in res], tv
        le, _ = micro_losses(False)
        lg, tvg = micro_losses(mode)
        assert len(tvg._graphs) == 1
        assert len({round(x, 6) for x in lg}) == 4, lg           # four different micro-batches, four different losses
        np.testing.assert_allclose(lg, le, rtol=0, atol=2e-5)
    finally:
        E.DROPOUT_KEEP = keep


def test_graph_cache_is_bounded(mode):
    """Variable-N sources: every distinct point count would otherwise pin a captured graph and its activations for ever."""
    import importlib
    TV = importlib.import_module("dgcnn.trainval")               # (the package attribute `dgcnn.trainval` is the class)
    tv = dgcnn.trainval(_flags(TRAIN=False)).initialize().use_graph(mode)
    old = TV.GRAPH_CACHE_MAX
    TV.GRAPH_CACHE_MAX = 3
    try:
        for n in (256, 288, 320, 352, 384):
            x = torch.rand(1, n, 3, device="cuda")
            for _ in range(3):
                out = tv.inference(None, [x])
            assert out[0].shape == (1, n, 2)
        assert len(tv._graphs) == 3
        assert [k[1][1] for k in tv._graphs] == [320, 352, 384]   # least recently used shapes were dropped
        if mode == "plan":                                        # ... and the bytes a plan's private pool pins are known (the second bound)
            assert all(e["pool_bytes"] > 0 for e in tv._graphs.values())
            oldf, TV.GRAPH_CACHE_MAX_FRACTION = TV.GRAPH_CACHE_MAX_FRACTION, 0.0
            try:
                x = torch.rand(1, 448, 3, device="cuda")
                for _ in range(3):
                    tv.inference(None, [x])
                assert len(tv._graphs) == 1 and next(iter(tv._graphs))[1][1] == 448      # a zero byte budget keeps only the newest plan
            finally:
                TV.GRAPH_CACHE_MAX_FRACTION = oldf
        tv.use_graph("auto")                                      # "auto": a shape must come back a few times first
        x = torch.rand(1, 416, 3, device="cuda")
        for i in range(TV.GRAPH_CAPTURE_AFTER_AUTO):
            tv.inference(None, [x])
            assert all(k[1][1] != 416 for k in tv._graphs)
        tv.inference(None, [x])
        assert any(k[1][1] == 416 and k[0] == "plan" for k in tv._graphs)          # "auto" replays from launch plans
    finally:
        TV.GRAPH_CACHE_MAX = old


def test_launch_plan_with_the_second_stream_retraces_the_eager_step_bit_for_bit():
    """The recorded plan contains the cross-stream waits of the step (weight-gradient GEMMs, transposed adjacency and the
    parameter-only preparation on the side stream): with the deterministic kernels a replayed training run must equal the eager
    run bit for bit, and the plan must hold more than one wait and no collective."""
    pts, lab = _data(steps=3, B=4, N=512, seed=5)
    old_min = E.SIDE_STREAM_MIN_ROWS
    E.SIDE_STREAM_MIN_ROWS = 0                                   # the test shape is below the production threshold
    try:
        tv_e, loss_e, p_e = _train(False, 3, pts, lab, DETERMINISTIC=True)
        tv_p, loss_p, p_p = _train("plan", 3, pts, lab, DETERMINISTIC=True)
        info = tv_p.launch_plan_info()
        assert len(info) == 1 and info[0]["kernels"] > 60 and info[0]["waits"] >= 4 and info[0]["collectives"] == 0, info
        np.testing.assert_allclose(loss_p, loss_e, rtol=0, atol=1e-6)      # (the REPORTED loss is summed with atomics: last bit)
        np.testing.assert_array_equal(p_p, p_e)                             # what training depends on: bit for bit
    finally:
        E.SIDE_STREAM_MIN_ROWS = old_min
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT


def test_replay_restages_inputs_that_changed_in_place_or_are_new_objects(mode):
    """A replayed tower reads its own input buffers; the caller's tensor is copied into them unless the SAME tensor object,
    unmodified, was staged last time.  In-place edits (also through a view) and new tensors at a recycled address must be seen."""
    f = _flags(TRAIN=False, STATIC_INPUTS=True)                  # the skip is opt-in; without the flag every replay stages
    tv = dgcnn.trainval(f).initialize()
    rng = np.random.default_rng(8)
    a = torch.from_numpy(rng.random((2, 384, 3), dtype=np.float32)).cuda()
    b = torch.from_numpy(rng.random((2, 384, 3), dtype=np.float32)).cuda()
    want_a = tv.inference(None, [a])[0].clone()
    want_b = tv.inference(None, [b])[0].clone()
    assert float((want_a - want_b).abs().max()) > 1e-3
    tv.use_graph(mode)
    x = a.clone()
    for _ in range(3):                                            # sighting, capture, replay (staged once, then skipped)
        got = tv.inference(None, [x])[0]
    np.testing.assert_allclose(got.cpu().numpy(), want_a.cpu().numpy(), rtol=0, atol=2e-4)
    x.view(-1)[:] = b.view(-1)                                    # in-place edit through a view: the version counter moves
    np.testing.assert_allclose(tv.inference(None, [x])[0].cpu().numpy(), want_b.cpu().numpy(), rtol=0, atol=2e-4)
    addr = x.data_ptr()
    del x, got
    y = a.clone()                                                 # a NEW tensor, very likely at the recycled address, version 0
    out = tv.inference(None, [y])[0]
    np.testing.assert_allclose(out.cpu().numpy(), want_a.cpu().numpy(), rtol=0, atol=2e-4)
    print("recycled address: %s" % (y.data_ptr() == addr))


def test_default_replay_stages_inputs_written_behind_torchs_back(mode):
    """Without STATIC_INPUTS a replay always copies the caller's tensors in: a write that does not move torch's version counter
    (here: one of the library's own kernels through data_ptr) must be seen."""
    from dgcnn import _hip as H
    tv = dgcnn.trainval(_flags(TRAIN=False)).initialize()
    rng = np.random.default_rng(9)
    a = torch.from_numpy(rng.random((2, 320, 3), dtype=np.float32)).cuda()
    b = torch.from_numpy(rng.random((2, 320, 3), dtype=np.float32)).cuda()
    want_b = tv.inference(None, [b])[0].clone()
    tv.use_graph(mode)
    x = a.clone()
    for _ in range(3):
        tv.inference(None, [x])
    v = x._version
    H.call("dgcnn_copy2d_f32", b.data_ptr(), 3, x.data_ptr(), 3, 2 * 320, 3, 0)       # x <- b, invisible to the version counter
    assert x._version == v
    got = tv.inference(None, [x])[0]
    np.testing.assert_allclose(got.cpu().numpy(), want_b.cpu().numpy(), rtol=0, atol=2e-4)


def test_capture_of_a_small_step_after_a_large_one_leaves_the_arena_zeroed(mode):
    """a recorded step zeroes only what IT uses of the statistics arena.  Eager large step -> capture of a small step ->
    eager large step: the last one must equal the same step in a run that never captured (deterministic kernels: bit for bit)."""
    rng = np.random.default_rng(21)
    big = torch.from_numpy(rng.random((6, 1024, 3), dtype=np.float32)).cuda()
    bigl = torch.from_numpy(rng.integers(0, 2, (6, 1024)).astype(np.int32)).cuda()
    small = torch.from_numpy(rng.random((1, 256, 3), dtype=np.float32)).cuda()
    smalll = torch.from_numpy(rng.integers(0, 2, (1, 256)).astype(np.int32)).cuda()
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    try:
        def run(graph):
            tv = dgcnn.trainval(_flags(DETERMINISTIC=True)).initialize().use_graph(False)
            c = dgcnn.ctx()
            tv.zero_gradients(None)
            tv.accum_gradient(None, [big], [bigl])                # eager, dirties a large extent of the arena
            big_off = c.stat_off
            tv.use_graph(graph)
            for _ in range(3):                                    # sighting, capture, replay of the SMALL inference step
                tv.inference(None, [small], [smalll])
            if graph:
                assert len(tv._graphs) == 1
            tv.use_graph(False)
            tv.zero_gradients(None)
            res = tv.accum_gradient(None, [big], [bigl])          # eager again: assumes everything behind stat_off is zero
            assert c.stat_off == big_off
            return float(res[2]), c.flat_grad.cpu().numpy().copy()
        l0, g0 = run(False)
        l1, g1 = run(mode)
        assert abs(l0 - l1) < 1e-6                                # (the REPORTED loss is summed with atomics: last bit)
        np.testing.assert_array_equal(g1, g0)
    finally:
        E.DROPOUT_KEEP = keep
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT


def test_a_recording_whose_zeroed_extent_is_too_small_is_discarded_not_fatal():
    """the arena extent a recording zeroes comes from an earlier eager sighting; if the step then uses more (slot count
    or arena changed in between), the run must go on: the recording step's own results are right (everything behind the recorded
    extent was zeroed before the recording), the plan is dropped, and the next recording zeroes the full extent."""
    pts, lab = _data(steps=1, B=4, N=512, seed=13)
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    try:
        def run(sabotage):
            tv = dgcnn.trainval(_flags(DETERMINISTIC=True)).initialize().use_graph("plan")
            c = dgcnn.ctx()
            grads = []
            for i in range(5):
                if sabotage and i == 1:
                    assert len(tv._graph_used) == 1                      # the sighting measured the step's extent ...
                    for k in tv._graph_used:
                        tv._graph_used[k] = 8                            # ... and something made it far too small
                tv.zero_gradients(None)
                tv.accum_gradient(None, [pts[0]], [lab[0]])
                grads.append(c.flat_grad.cpu().numpy().copy())
            return tv, grads
        tv0, g0 = run(False)
        tv1, g1 = run(True)
        for a, b in zip(g0, g1):
            np.testing.assert_array_equal(a, b)                          # every step, the discarded recording's included
        assert len(tv1._graphs) == 1                                     # a later recording (full extent) took over
    finally:
        E.DROPOUT_KEEP = keep
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT
