"""Step executors and stream ordering (round-2 fixes):
  * a recorded launch list replays bitwise what eager launches compute (dropout on, two streams, Adam);
  * the side-stream -> Adam edge holds even when the side stream is artificially late;
  * LR schedules reach the Adam kernel under hipGraph and launch-list replay (hyper-parameters live on the device);
  * ragged shape sequences never free buffers that recordings point to (grow-only allocations);
  * weight decay leaves parameters outside the caption path alone."""
import pytest
import torch

from test_dist_gpu import MC, VOCAB

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _model(dtype=torch.bfloat16, dropout=0.3, seed=7):
    from helpers import build_model
    torch.manual_seed(seed)
    mc = dict(MC)
    mc["dropout"] = dropout
    m = build_model(mc, VOCAB, DEV, dtype)
    m.train()
    return m


def _batch(seed, B=6, T=7, S=9):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 48, generator=g)
    mask = torch.zeros(B, T, dtype=torch.bool); mask[1, T - 2:] = True
    ids = torch.randint(3, VOCAB, (B, S), generator=g); ids[:, 0] = 101; ids[2, S - 3:] = 0
    return feats.to(DEV), mask.to(DEV), ids.to(DEV)


def _run(executor, steps=5, shapes=None, lr_of=None, dtype=torch.bfloat16):
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    m = _model(dtype)
    m._seed.fill_(1234)
    opt = FusedAdam(m, lr=1e-3)
    tr = CaptionTrainer(m, opt, use_graph=executor == "graph", launch_list=executor == "list")
    losses = []
    for k in range(steps):
        if lr_of is not None:
            opt.param_groups[0]["lr"] = lr_of(k)
        kw = {} if shapes is None else dict(zip(("B", "T", "S"), shapes[k % len(shapes)]))
        losses.append(tr.step(*_batch(100 + k, **kw)).clone())
    torch.cuda.synchronize()
    return m.flat_params.clone(), torch.cat(losses), tr


@pytest.mark.parametrize("executor", ["list", "graph"])
def test_recorded_step_is_bitwise_the_eager_step(executor):
    p0, l0, _ = _run("eager")
    p1, l1, tr = _run(executor)
    assert torch.equal(l0, l1)
    assert torch.equal(p0, p1)
    if executor == "list":
        (ll, _loss), = tr._lists.values()
        assert len(ll) > 50 and ll.n_streams == 2       # both streams and their edges were recorded


@pytest.mark.parametrize("executor", ["eager", "list", "graph"])
def test_lr_schedule_reaches_the_adam_kernel(executor):
    """A different learning rate every step: replays must follow it (kernel scalars are frozen at record time)."""
    sched = lambda k: 1e-3 * (0.5 ** k)
    p_ref, _, _ = _run("eager", lr_of=sched)
    p_const, _, _ = _run("eager")
    p, _, _ = _run(executor, lr_of=sched)
    assert torch.equal(p, p_ref)
    assert not torch.equal(p_ref, p_const)


def test_side_stream_gradients_are_final_before_adam_reads_them(monkeypatch):
    """Delay the side stream (decoder weight gradients, d(memory) GEMMs) by ~20 ms each step: Adam on the main stream
    must still see final gradients, i.e. the result equals the run without the delay."""
    from vct_amd import engine
    ref, lref, _ = _run("eager", steps=3)
    orig = engine._StackBase.flush_dw
    spins = {"n": 0}

    def slow_flush(self, main=False):
        if not main and self._dw_pending and self.side is not None:
            with torch.cuda.stream(self.side):
                torch.cuda._sleep(40_000_000)
            spins["n"] += 1
        return orig(self, main)
    monkeypatch.setattr(engine._StackBase, "flush_dw", slow_flush)
    got, lgot, _ = _run("eager", steps=3)
    assert spins["n"] > 0
    assert torch.equal(lref, lgot)
    assert torch.equal(ref, got)


@pytest.mark.parametrize("executor", ["list", "graph"])
def test_ragged_epoch_keeps_recordings_valid(executor):
    """Eight shape configurations cycled twice (the loader trims S per batch): recordings of earlier shapes must stay
    valid -- buffers only grow, and a growth drops every recording that baked the old pointers."""
    shapes = [(6, 7, 9), (6, 7, 5), (4, 7, 8), (6, 5, 9), (6, 7, 7), (3, 6, 6), (6, 7, 10), (5, 7, 9)]
    p0, l0, _ = _run("eager", steps=16, shapes=shapes)
    p1, l1, tr = _run(executor, steps=16, shapes=shapes)
    assert torch.equal(l0, l1)
    assert torch.equal(p0, p1)
    assert len(tr._lists if executor == "list" else tr._graphs) >= 2


def test_weight_decay_spares_parameters_outside_the_caption_path():
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    m = _model(torch.float32, dropout=0.0)
    assert m.caption_param_end < m._ps.total                 # matching.v_proj.* sit behind the caption parameters
    before = m.flat_params.clone()
    opt = FusedAdam(m, lr=1e-2, weight_decay=0.1)
    CaptionTrainer(m, opt).step(*_batch(3))
    torch.cuda.synchronize()
    e = m.caption_param_end
    assert torch.equal(m.flat_params[e:], before[e:])
    assert not torch.equal(m.flat_params[:e], before[:e])


def test_taps_time_kernels_inside_replays():
    from vct_amd import ops
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    m = _model()
    tr = CaptionTrainer(m, FusedAdam(m, lr=1e-3), launch_list=True)
    ops.taps_enable(True)
    try:
        for k in range(4):
            tr.step(*_batch(5))
        torch.cuda.synchronize()
        ms = ops.tap_collect("step")
        gen = ops.tap_collect("gen_fwd")
    finally:
        ops.taps_enable(False)
    assert len(ms) == 4 and all(0.0 < x < 1000.0 for x in ms)
    assert len(gen) == 4 and all(0.0 < x <= y for x, y in zip(gen, ms))
    # a restricted bracket set (what bench.py keeps inside its timed region); recordings made under another set are dropped
    for tag in ops.TAPS:
        ops.tap_collect(tag)                                  # drain the brackets of the first phase
    ops.taps_enable(True, only=("gen_fwd",))
    try:
        tr.drop_recordings()
        for k in range(3):
            tr.step(*_batch(5))
        torch.cuda.synchronize()
        assert len(ops.tap_collect("gen_fwd")) == 3 and ops.tap_collect("step") == [] and ops.tap_collect("adam") == []
    finally:
        ops.taps_enable(False)


@pytest.mark.parametrize("sharded", [True, False])
@pytest.mark.parametrize("executor", ["eager", "list"])
def test_own_rccl_communicator_world1_step_is_the_plain_step(sharded, executor):
    """The library's RCCL communicator through the C ABI (vct_comm_*), world size 1 (the test box has one GPU): reduce-scatter ->
    Adam on the owned shard -> all-gather (or all-reduce + Adam) per bucket on the communicator's stream must give bitwise
    the parameters of the plain single-GPU step -- eagerly and as a recorded launch list."""
    from vct_amd.comm import RcclColl
    from vct_amd.trainer import CaptionTrainer, FusedAdam, ShardedExchange
    ref, lref, _ = _run("eager", steps=4)
    m = _model()
    m._seed.fill_(1234)
    opt = FusedAdam(m, lr=1e-3)
    coll = RcclColl(device=torch.device("cuda", 0))
    assert coll.world == 1 and coll.self_test()
    ex = ShardedExchange(m, opt, coll, sharded=sharded)
    tr = CaptionTrainer(m, opt, ex, launch_list=executor == "list")
    assert tr.use_list == (executor == "list")
    losses = torch.cat([tr.step(*_batch(100 + k)).clone() for k in range(4)])
    torch.cuda.synchronize()
    assert torch.equal(losses, lref)
    assert torch.equal(m.flat_params, ref)
    assert torch.equal(m._ps.cflat[:opt.skip[0]], m.flat_params[:opt.skip[0]].to(torch.bfloat16))
    coll.close()


def test_rccl_bf16_payload_world1_is_close_to_the_fp32_step():
    from vct_amd.comm import RcclColl
    from vct_amd.trainer import CaptionTrainer, FusedAdam, ShardedExchange
    ref, _, _ = _run("eager", steps=2)
    m = _model()
    m._seed.fill_(1234)
    opt = FusedAdam(m, lr=1e-3)
    coll = RcclColl(device=torch.device("cuda", 0))
    tr = CaptionTrainer(m, opt, ShardedExchange(m, opt, coll, payload_dtype=torch.bfloat16))
    for k in range(2):
        tr.step(*_batch(100 + k))
    torch.cuda.synchronize()
    e = m.caption_param_end
    assert float((m.flat_params[:e] - ref[:e]).abs().max()) < 5e-3      # Adam normalises: a bf16-rounded gradient moves a weight <= 2 lr
    coll.close()


@pytest.mark.parametrize("executor", ["eager", "list"])
def test_adam_after_the_joined_backward_is_the_same_step(executor, monkeypatch):
    """A/B switch CaptionTrainer.adam_after_backward: the whole Adam pass behind the joined backward instead of most of it beside the
    encoder backward -- a different stream schedule of the same kernels: parameters and losses must be bitwise equal."""
    from vct_amd.trainer import CaptionTrainer
    p0, l0, _ = _run(executor)
    monkeypatch.setattr(CaptionTrainer, "adam_after_backward", True)
    p1, l1, _ = _run(executor)
    assert torch.equal(l0, l1) and torch.equal(p0, p1)


@pytest.mark.parametrize("executor", ["eager", "list"])
def test_optimizer_inside_the_weight_gradient_gemms_is_the_separate_pass(executor, monkeypatch):
    """CaptionTrainer.fuse_adam (default on one GPU): every weight matrix is stepped in the epilogue of its own weight-gradient GEMM,
    the rest by one multi-range launch.  Same kernels otherwise, same arithmetic: parameters, both moments and the bf16 shadow
    must be BITWISE what the separate optimizer pass leaves (the vocabulary dX in its NN form on both sides: the epilogue
    maintains no W_g^T)."""
    from vct_amd.engine import DecoderEngine
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    monkeypatch.setattr(DecoderEngine, "gen_dx_nt", False)
    res = {}
    for fuse in (True, False):
        m = _model()
        m._seed.fill_(1234)
        opt = FusedAdam(m, lr=1e-3, weight_decay=0.01)
        tr = CaptionTrainer(m, opt, launch_list=executor == "list")
        assert tr.fuse_adam is True
        tr.fuse_adam = fuse
        losses = torch.cat([tr.step(*_batch(100 + k)).clone() for k in range(4)])
        torch.cuda.synchronize()
        e = m.caption_param_end
        res[fuse] = (losses, m.flat_params.clone(), opt.exp_avg[:e].clone(), opt.exp_avg_sq[:e].clone(), m._ps.cflat[:e].clone(),
                     int(opt.step_dev))
        assert m._ps.dw_adam is None                         # the hook is installed only while the trainer enqueues a step
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
    assert res[True][5] == 4


def test_plain_backward_beside_a_fusing_trainer_still_produces_gradients():
    """The reference call sequence (train.py:123-125) on a model whose trainer fuses the optimizer into the weight-gradient GEMMs:
    loss.backward() outside the trainer leaves the parameters alone and fills .grad."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    m = _model(dropout=0.0)
    tr = CaptionTrainer(m, FusedAdam(m, lr=1e-3), launch_list=True)
    tr.step(*_batch(100)); tr.step(*_batch(101))
    torch.cuda.synchronize()
    before = m.flat_params.clone()
    f, k, i = _batch(102)
    m.zero_grad(set_to_none=False)
    m([f], [k], i).backward()
    torch.cuda.synchronize()
    assert torch.equal(m.flat_params, before)
    g = m._ps.g["cap_decoder.generator.weight"]
    assert float(g.abs().max()) > 0 and bool(torch.isfinite(g).all())


@pytest.mark.parametrize("executor", ["eager", "list"])
def test_weight_gradients_of_a_fused_step_are_flagged_stale_or_kept(executor):
    """The reference leaves valid .grad after backward (train.py:125).  The single-GPU bf16 trainer consumes the weight-matrix
    gradients inside the weight-gradient GEMMs: model.grads_valid is False after such a step and check_grads_valid() raises;
    CaptionTrainer(keep_weight_grads=True) stores them as well -- same parameters bit for bit, and the stored gradients are the
    gradients a plain backward of the same batch on the same weights produces."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    res = {}
    for keep in (False, True):
        m = _model(dropout=0.0)
        opt = FusedAdam(m, lr=1e-3)
        tr = CaptionTrainer(m, opt, launch_list=executor == "list", keep_weight_grads=keep)
        assert m.grads_valid is True
        before = m.flat_params.clone()
        tr.step(*_batch(100))
        torch.cuda.synchronize()
        assert m.grads_valid is keep
        if not keep:
            with pytest.raises(RuntimeError):
                m.check_grads_valid()
        else:
            m.check_grads_valid()
            g_step = m.flat_grads.clone()
            # the same gradients from a plain backward at the weights the step started from
            m2 = _model(dropout=0.0)
            m2.flat_params.copy_(before); m2._ps.refresh_shadow(force=True)
            f, k, i = _batch(100)
            m2.zero_grad(set_to_none=False)
            m2([f], [k], i).backward()
            torch.cuda.synchronize()
            assert m2.grads_valid is True
            for name in ("cap_decoder.generator.weight", "cap_decoder.decoder.layers.0.linear1.weight",
                         "video_encoder.transformer_encoder.layers.1.self_attn.in_proj_weight", "cap_decoder.generator.bias"):
                assert torch.equal(m._ps.g[name], m2._ps.g[name]), name
            assert float(g_step.abs().max()) > 0
        tr.step(*_batch(101))
        torch.cuda.synchronize()
        res[keep] = m.flat_params.clone()
    assert torch.equal(res[False], res[True])


def test_adopted_input_buffers_skip_the_staging_copies_and_give_the_same_step():
    """CaptionTrainer.adopt_inputs: a producer that writes its batches into the executor's own static buffers; same losses and
    parameters as passing fresh tensors (which step() copies into those buffers)."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    p0, l0, _ = _run("list", steps=4)
    m = _model()
    m._seed.fill_(1234)
    opt = FusedAdam(m, lr=1e-3)
    tr = CaptionTrainer(m, opt, launch_list=True)
    f, k, i = tr.adopt_inputs(*_batch(100))
    losses = []
    for s in range(4):
        nf, nk, ni = _batch(100 + s)
        f.copy_(nf); k.copy_(nk); i.copy_(ni)               # the producer fills the adopted buffers in place
        losses.append(tr.step(f, k, i).clone())
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(losses), l0) and torch.equal(m.flat_params, p0)


def test_vocabulary_dx_through_the_maintained_transposed_shadow(monkeypatch):
    """DecoderEngine.gen_dx_nt: dX = dlogits W_g in the NT form reads W_g^T, which ParamSet keeps beside the bf16 shadow (rewritten
    behind every optimizer pass over W_g).  Same training trajectory as the NN form up to bf16 GEMM rounding, and W_g^T equals the
    shadow's transpose after every step (eager and recorded)."""
    from vct_amd.engine import DecoderEngine
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    p0, l0, _ = _run("list", steps=4)
    monkeypatch.setattr(DecoderEngine, "gen_dx_nt", True)
    for executor in ("eager", "list"):
        m = _model()
        m._seed.fill_(1234)
        tr = CaptionTrainer(m, FusedAdam(m, lr=1e-3), launch_list=executor == "list")
        tr.fuse_adam = False         # (the optimizer epilogue of the weight-gradient GEMMs keeps no W_g^T: this is the separate-pass configuration)
        losses = [tr.step(*_batch(100 + k)).clone() for k in range(4)]
        torch.cuda.synchronize()
        ps = m._ps
        name = "cap_decoder.generator.weight"
        assert name in ps.transposed
        t = ps.transposed[name][0]
        w = ps.c[name]
        assert torch.equal(t[:, :w.shape[0]], w.t())
        l1 = torch.cat(losses)
        assert float((l1 - l0).abs().max()) < 2e-2 * float(l0.abs().max())
        assert float((m.flat_params - p0).abs().max()) < 5e-3


def test_host_commands_run_in_launch_order_on_replay():
    """ops.host_call (vct_cmdlist_host_call): host work recorded into a launch list runs at its place of every replay, after the
    stream work enqueued before it has finished (it reads a value a preceding kernel wrote), and its failure is the replay's
    status.  This is how host-side collectives (gloo) ride in a recorded step; RCCL collectives are plain stream work."""
    from vct_amd import ops
    src = torch.arange(8, dtype=torch.float32, device=DEV)
    dst = torch.zeros(8, dtype=torch.bfloat16, device=DEV)
    seen = []
    ll = ops.LaunchList()
    with ll.record():
        ops.cast(src, dst)
        ops.host_call(lambda: seen.append(dst.float().sum().item()))
    assert seen == [] and len(ll) == 2
    ll.replay()
    src.mul_(2)
    ll.replay()
    torch.cuda.synchronize()
    assert seen == [28.0, 56.0]
    boom = ops.LaunchList()
    with boom.record():
        ops.host_call(lambda: 1 / 0)
    with pytest.raises(RuntimeError):
        boom.replay()


def test_two_models_in_one_process_keep_their_own_recordings():
    """A training model and an evaluation copy in one process (train.py:244-249 validates between epochs): each has its own side
    stream and buffer generation (engine.StepContext) -- the copy's buffer growth must not drop the trainer's recordings, and
    interleaving the two must not change either one's results."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    ref, lref, _ = _run("list", steps=6)

    a = _model()
    a._seed.fill_(1234)
    opt = FusedAdam(a, lr=1e-3)
    tr = CaptionTrainer(a, opt, launch_list=True)
    b = _model(seed=11)                       # the "evaluation copy": different weights, its own parameter set
    b.eval()
    assert a._ps.ctx is not b._ps.ctx
    losses, lists_seen = [], []
    for k in range(6):
        losses.append(tr.step(*_batch(100 + k)).clone())
        lists_seen.append(id(next(iter(tr._lists.values()))[0]))
        # grow the copy's buffers (a bigger batch each time) and run a recorded evaluation step + a captured decode in between
        fb, mb_, ib = _batch(500 + k, B=6 + 2 * k, T=7 + k, S=9 + k)
        with torch.no_grad():
            lb = b._forward_loss(fb, mb_, ib, False)[0]
        b.greedy_decode_ids([fb], None, max_len=6)
        assert bool(torch.isfinite(lb).all())
    torch.cuda.synchronize()
    assert a._ps.ctx.side is not None and a._ps.ctx.side != b._ps.ctx.side
    assert len(set(lists_seen)) == 1, "the copy's buffer growth dropped the trainer's recording"
    assert torch.equal(torch.cat(losses), lref) and torch.equal(a.flat_params, ref)


def test_two_models_training_alternately_equal_their_solo_runs():
    """Two models TRAINING in one process, one recorded step each in turn: neither may see the other's state -- in particular the
    token-embedding gradient of the single-GPU fast path re-zeroes only the rows of the ids its own previous step used, which a
    workspace shared between the models got wrong (a replay zeroed the rows of the OTHER model's batch)."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    solo = {}
    for seed in (7, 11):
        m = _model(seed=seed)
        m._seed.fill_(1234 + seed)
        tr = CaptionTrainer(m, FusedAdam(m, lr=1e-3), launch_list=True)
        for k in range(5):
            tr.step(*_batch(100 + k + seed))
        torch.cuda.synchronize()
        solo[seed] = m.flat_params.clone()
    ms, trs = {}, {}
    for seed in (7, 11):
        ms[seed] = _model(seed=seed)
        ms[seed]._seed.fill_(1234 + seed)
        trs[seed] = CaptionTrainer(ms[seed], FusedAdam(ms[seed], lr=1e-3), launch_list=True)
    for k in range(5):
        for seed in (7, 11):
            trs[seed].step(*_batch(100 + k + seed))
    torch.cuda.synchronize()
    for seed in (7, 11):
        assert torch.equal(ms[seed].flat_params, solo[seed]), seed


def test_exclusive_gradient_zeroing_is_per_engine():
    """Trainer A (single GPU, fused optimizer) makes ITS decoder engine the exclusive writer of its gradient buffer, so its
    token-embedding gradient re-zeroes only the rows of its previous batch.  A second model B in the same process whose
    gradient table is also written by somebody else (in-place accumulation / averaging between backward calls) must keep the
    full zero-fill: the flag is an attribute of A's engine, not of the class."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    solo = _model(seed=7); solo._seed.fill_(77)
    tr = CaptionTrainer(solo, FusedAdam(solo, lr=1e-3), launch_list=True)
    for k in range(4):
        tr.step(*_batch(100 + k))
    torch.cuda.synchronize()
    a = _model(seed=7); a._seed.fill_(77)
    b = _model(seed=11, dropout=0.0)
    tra = CaptionTrainer(a, FusedAdam(a, lr=1e-3), launch_list=True)
    assert a.cap_decoder._engine().exclusive_grads is True and b.cap_decoder._engine().exclusive_grads is False
    key = "cap_decoder.tgt_to_emb.weight"
    grads = []
    for k in range(4):
        tra.step(*_batch(100 + k))
        fb, mb, ib = _batch(900)                       # B: the SAME batch every time, reference call sequence (train.py:123-125)
        b.zero_grad(set_to_none=False)
        loss = b([fb], [mb], ib)
        loss.backward()
        grads.append(b._ps.g[key].clone())
        b._ps.g[key].add_(1.0)                         # somebody else writes B's table (every row, also rows no batch touches)
    torch.cuda.synchronize()
    assert torch.equal(a.flat_params, solo.flat_params)
    for g in grads[1:]:
        assert torch.equal(g, grads[0])                # B's backward re-zeroed ALL rows each time
    used = torch.zeros(VOCAB, dtype=torch.bool, device=DEV); used[_batch(900)[2][:, :-1].reshape(-1)] = True
    assert float(grads[0][~used].abs().max()) == 0.0


@pytest.mark.parametrize("executor", ["list", "graph"])
def test_replay_follows_weights_loaded_outside_the_optimizer(executor):
    """load_state_dict between two replayed steps (restoring the best checkpoint, train.py:214-216): the replayed step must
    compute with the NEW weights -- the bf16 shadow the GEMMs read is re-cast before the replay."""
    from vct_amd.trainer import CaptionTrainer, FusedAdam

    def run(ex):
        m = _model()
        m._seed.fill_(99)
        other = {k: v.detach().clone() for k, v in _model(seed=21).state_dict().items()}
        opt = FusedAdam(m, lr=1e-3)
        tr = CaptionTrainer(m, opt, use_graph=ex == "graph", launch_list=ex == "list")
        out = []
        for k in range(5):
            if k == 3:
                m.load_state_dict(other)
            out.append(tr.step(*_batch(100 + k)).clone())
        torch.cuda.synchronize()
        return torch.cat(out), m.flat_params.clone()
    l0, p0 = run("eager")
    l1, p1 = run(executor)
    assert torch.equal(l0, l1) and torch.equal(p0, p1)


def test_decode_after_training_steps_uses_the_current_weights():
    """The batch-1 decode kernels read TRANSPOSED copies of out_proj / linear2 that are refreshed on demand (not by every optimizer
    step): decoding, training a few recorded steps, decoding again must equal a fresh model loaded with the trained weights --
    with the captured token-step graphs of the first call still in use."""
    from helpers import build_model
    from vct_amd.trainer import CaptionTrainer, FusedAdam
    mc = dict(MC, embed_dim=512, modal_shape=[48])
    mc["video_encoder"] = dict(mc["video_encoder"], layer=1, nhead=8, feedforward=256)
    mc["caption_decoder"] = dict(mc["caption_decoder"], layer=2, nhead=8, feedforward=256)
    torch.manual_seed(5)
    m = build_model(mc, VOCAB, DEV, torch.bfloat16)
    feats = torch.randn(1, 7, 48, generator=torch.Generator().manual_seed(1)).to(DEV)
    from vct_amd import engine
    ys0 = m.greedy_decode_ids([feats], None, max_len=8)
    st = next(iter(m.__dict__["_decode_sessions"].values()))
    assert engine._decoder_block_decode_ok(m.cap_decoder._engine(), st)          # the transposed-weight path is the one in use
    m.train()
    tr = CaptionTrainer(m, FusedAdam(m, lr=5e-2), launch_list=True)
    for k in range(4):
        tr.step(*_batch(300 + k))
    torch.cuda.synchronize()
    ys1 = m.greedy_decode_ids([feats], None, max_len=8)
    ref = build_model(mc, VOCAB, DEV, torch.bfloat16)
    ref.load_state_dict(m.state_dict())
    ys_ref = ref.greedy_decode_ids([feats], None, max_len=8)
    assert torch.equal(ys1, ys_ref)
    for name, ent in m._ps.transposed.items():
        if not ent[3]:
            w = m._ps.c[name]
            assert torch.equal(ent[0][:, :w.shape[0]], w.t()), name
