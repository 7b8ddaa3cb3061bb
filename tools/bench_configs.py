"""Device-time measurements of the other benchmark configurations (the headline one is bench.py):
cfg 2 = 1024 x 2^16 forward + inverse, cfg 3 = one 2^24 transform (single call and the sharded code
path on one rank), plus 2^20 inverse. Prints one JSON object."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stark_verifier_amd as sv
from stark_verifier_amd import fft
from stark_verifier_amd.sharding import ntt_sharded, HipLocalOps

P = 2**256 - 351 * 2**32 + 1
ctx = sv.Context(0)
ts = torch.cuda.ExternalStream(ctx.stream)


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ctx.synchronize(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ts):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
    ctx.synchronize(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def fill(count, seed):
    t = torch.empty((count, 8), dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.stark_fill_random_field_dev(ctx.handle, t.data_ptr(), count, seed, None))
    ctx.synchronize()
    return t


out = {}
for name, log_n, B in (("cfg2_2^16_x1024", 16, 1024), ("metric_2^20_x32", 20, 32), ("cfg3_2^24_x1", 24, 1), ("2^12_x16384", 12, 16384), ("2^10_x65536", 10, 65536)):
    n = 1 << log_n
    w = pow(7, (P - 1) >> log_n, P)
    x = fill(B * n, 5 + log_n)
    y = torch.empty_like(x)
    f = timed(lambda: fft.ntt_batch_dev(x.data_ptr(), y.data_ptr(), log_n, B, w, False, ctx))
    i = timed(lambda: fft.ntt_batch_dev(y.data_ptr(), x.data_ptr(), log_n, B, w, True, ctx))
    bf = (n // 2) * log_n * B
    out[name] = {"forward_ms": f, "inverse_ms": i, "forward_transforms_per_s": B / f * 1e3, "inverse_transforms_per_s": B / i * 1e3,
                 "forward_Gbutterflies_per_s": bf / f * 1e-6, "GB_moved_compulsory_per_s": 2 * 32 * n * B / f * 1e-6}
    if log_n == 24:
        ops = HipLocalOps(ctx)
        s = timed(lambda: ntt_sharded(x, log_n, w, False, ops=ops), reps=3, warm=1)
        s1 = timed(lambda: ntt_sharded(x, log_n, w, False, ops=ops, in_layout="strided", out_layout="strided"), reps=3, warm=1)
        out[name]["sharded_path_one_rank_blocked_ms"] = s
        out[name]["sharded_path_one_rank_strided_ms"] = s1
        out[name]["sharded_path_one_rank_rows_to_columns_ms"] = timed(
            lambda: ntt_sharded(x, log_n, w, False, ops=ops, in_layout="strided", out_layout="columns"), reps=3, warm=1)
        out[name]["sharded_three_step_one_rank_strided_ms"] = timed(
            lambda: ntt_sharded(x, log_n, w, False, ops=ops, in_layout="strided", out_layout="strided", fused=False), reps=3, warm=1)
        # the two stages alone through the C entry points (what a non-Python host calls around its exchange)
        r = fft._limbs(w)
        z = torch.empty_like(x)
        out[name]["stage_a_ms"] = timed(lambda: ctx.check(ctx.lib.stark_ntt_sharded_stage_a_dev(
            ctx.handle, x.data_ptr(), z.data_ptr(), log_n, 1, 0, fft._ptr(r), 0, None)), reps=5)
        out[name]["stage_c_ms"] = timed(lambda: ctx.check(ctx.lib.stark_ntt_sharded_stage_c_dev(
            ctx.handle, z.data_ptr(), y.data_ptr(), log_n, 1, fft._ptr(r), 0, None)), reps=5)
        for world in (8,):
            out[name]["stage_a_ms_rank0_of_%d" % world] = timed(lambda: ctx.check(ctx.lib.stark_ntt_sharded_stage_a_dev(
                ctx.handle, x.data_ptr(), z.data_ptr(), log_n, world, 0, fft._ptr(r), 0, None)), reps=5)
            out[name]["stage_c_ms_rank0_of_%d" % world] = timed(lambda: ctx.check(ctx.lib.stark_ntt_sharded_stage_c_dev(
                ctx.handle, z.data_ptr(), y.data_ptr(), log_n, world, fft._ptr(r), 0, None)), reps=5)
    del x, y
print(json.dumps(out, indent=1))
