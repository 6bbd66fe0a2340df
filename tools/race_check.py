"""Run-to-run determinism of the product on one GPU: the same seeded step REPS times, forward outputs compared BITWISE with
the first run (the forward kernels have no atomics on data: any difference is a race), gradients compared with a tolerance
(BatchNorm sums and split-K weight gradients are accumulated with atomics: 1e-6-level jitter is expected, more is a race).

    python tools/race_check.py [arch] [B] [size] [dtype] [reps]      e.g.  resnet50 8 64 bf16 20
    MSFWSI_LIB=ab/lib_x.so python tools/race_check.py ...            another build of the library
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import build_product, flat_outputs, reference_loop_loss  # noqa: E402


def poison(gib=6):
    """dirty the caching allocator's free blocks with NaN bit patterns (0xff bytes: NaN as fp32, bf16 and fp16): a kernel
    that reads memory nobody wrote -- padding assumed zero, a tile edge -- then shows up as NaN / a changed result"""
    blocks = []
    for sz in (1 << 30, 1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12):
        n = max(4, min(256, (gib << 30) // 10 // sz))
        blocks += [torch.full((sz,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    del blocks


def trainer_mode(arch, B, size, dtype, reps, mode):
    """the FUSED trainer: `reps` times a freshly built, identically seeded model + PretrainStep run 4 steps (the first on
    one stream, the others on three, the inter_ group's Adam pass under the next step's encoder passes), free blocks
    poisoned between steps; the 4 losses and every updated weight are compared BITWISE with the first repetition.  Run it
    with MSFWSI_TUNING=15=1 (the reproducible mode): then any difference is a race between streams -- a missing
    event, a block handed to another stream too early -- not the order of atomic additions."""
    from msf_wsi_amd.train import PretrainStep
    from oracle import msfwsi_oracle as orc

    first, bad = None, 0
    for r in range(reps):
        model = build_product(arch, residual_gain=0.1).cuda().train()
        ts = PretrainStep(model, lr=1e-3, global_batch=B, dtype=dtype, arch=arch)
        losses, plans = [], []
        for t in range(4):
            (c1, c2), (t1, t2), idx = orc.make_batch("diverse", B, size, 16, t)
            losses.append(ts.step(((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx)))
            plans.append(ts.engine.last_plan)
            if "poison" in mode:
                torch.cuda.synchronize()
                poison(2)
        torch.cuda.synchronize()
        cur = {"loss": torch.stack(losses).cpu().clone()}
        cur.update({n: p.detach().float().cpu().clone() for n, p in model.named_parameters()})
        if r == 0:
            first = cur
            print(f"  plans of the 4 steps: {plans}; losses {[round(float(x), 6) for x in cur['loss'].ravel()]}", flush=True)
        else:
            # weights: bitwise.  The loss is an fp64 accumulator filled by atomic additions of 24 cosine kernels: their
            # order moves it by ~1e-16 relative (measured), which no weight ever sees
            diff = [k for k in cur if k != "loss" and not torch.equal(cur[k], first[k])]
            if not torch.allclose(cur["loss"], first["loss"], rtol=1e-12, atol=1e-15):
                diff.append("loss")
            if diff or not torch.isfinite(cur["loss"]).all():
                bad += 1
                k = diff[0] if diff else "loss"
                d = float((cur[k].double() - first[k].double()).norm() / (first[k].double().norm() + 1e-30))
                print(f"  repetition {r}: {len(diff)}/{len(cur)} tensors differ from repetition 0; first {k}: rel {d:.2e}", flush=True)
        del ts, model
    print(f"[race_check trainer {arch} B={B} size={size} {mode} splits-cap={os.environ.get('MSFWSI_TUNING', 'none')}] "
          f"{reps} repetitions of 4 steps: {bad} not bitwise equal to the first", flush=True)
    return bad


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dname = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    mode = sys.argv[6] if len(sys.argv) > 6 else "train"  # "train" | "nograd" (forward only under torch.no_grad) | "trainer" (fused multi-stream step) | +"-poison"
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dname]
    from oracle import msfwsi_oracle as orc  # inputs only (this is a test tool)

    if mode.startswith("trainer"):
        sys.exit(1 if trainer_mode(arch, B, size, dtype, reps, mode) else 0)

    (c1, c2), (t1, t2), idx = orc.make_batch("diverse", B, size, 16, 0)
    model = build_product(arch, residual_gain=0.1).cuda().train()
    if hasattr(model, "set_compute_dtype"):
        model.set_compute_dtype(dtype)
    args = ((c1.cuda(), t1.cuda()), (c2.cuda(), t2.cuda()), idx)
    first_out, first_grad = None, None
    bad_fwd, worst = 0, {}
    for r in range(reps):
        for p in model.parameters():
            p.grad = None
        if "poison" in mode and r > 0:
            poison()
        with torch.set_grad_enabled("nograd" not in mode), torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            outs = model(*args)
        loss, _ = reference_loop_loss(outs)
        if "nograd" not in mode:
            loss.backward()
        torch.cuda.synchronize()
        fo = {k: v.detach().float().clone() for k, v in flat_outputs(outs).items()}
        gr = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        if not all(bool(torch.isfinite(v).all()) for v in list(fo.values()) + list(gr.values())):
            print(f"  run {r}: NON-FINITE values: outputs " + str([k for k, v in fo.items() if not torch.isfinite(v).all()][:4])
                  + " grads " + str([k for k, v in gr.items() if not torch.isfinite(v).all()][:6]), flush=True)
        if first_out is None:
            first_out, first_grad = fo, gr
            continue
        diff = [k for k in fo if not torch.equal(fo[k], first_out[k])]
        if diff:
            bad_fwd += 1
            k = diff[0]
            d = float((fo[k] - first_out[k]).norm() / first_out[k].norm())
            print(f"  run {r}: {len(diff)}/{len(fo)} forward outputs differ from run 0; first {k}: rel {d:.2e}", flush=True)
        for n in gr:
            d = float((gr[n] - first_grad[n]).norm() / (first_grad[n].norm() + 1e-30))
            worst[n] = max(worst.get(n, 0.0), d)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(f"[race_check {arch} B={B} size={size} {dname} {mode} lib={os.environ.get('MSFWSI_LIB', 'default')}] {reps} runs: "
          f"{bad_fwd} with forward outputs not bitwise equal; gradient jitter worst " +
          ", ".join(f"{n} {d:.1e}" for n, d in top), flush=True)


if __name__ == "__main__":
    main()
