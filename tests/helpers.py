"""Shared test plumbing: seeded construction (same protocol as tests/golden/make_golden.py), golden loading."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
MODEL_SEED, HUB_SEED, DATA_SEED = 3407, 1234, 0
WEIGHTS = (0.1, 0.4, 0.7, 1.0)
LR = 1e-3


def install_hub_stub(residual_gain=1.0):
    """offline torch.hub stand-in: an un-pretrained net of the same arch under HUB_SEED, built with the
    product's own factories; the caller's RNG stream is untouched (same protocol as make_golden.py).
    residual_gain: scale of every residual branch's closing BatchNorm gain (make_golden.hub_stub: the well-conditioned
    cases stand in for the reference's TRAINED ImageNet weights with gain 0.1; manifest key stub_residual_gain)."""
    from msf_wsi_amd.models import resnet as my_resnet

    def fake(url, progress=True, **kw):
        arch = [k for k, v in my_resnet.model_urls.items() if v == url][0]
        state = torch.random.get_rng_state()
        torch.manual_seed(HUB_SEED)
        sd = my_resnet.__dict__[arch](pretrained=False).state_dict()
        torch.random.set_rng_state(state)
        if residual_gain != 1.0:
            last = ".bn2.weight" if arch in ("resnet18", "resnet34") else ".bn3.weight"
            for k in sd:
                if k.startswith("layer") and k.endswith(last):
                    sd[k] = sd[k] * residual_gain
        return sd

    torch.hub.load_state_dict_from_url = fake


def build_product(arch="resnet18", scale=4, residual_gain=1.0):
    from msf_wsi_amd.models import resnet as my_resnet
    from msf_wsi_amd.models.backbone import MSFWSI

    install_hub_stub(residual_gain)
    torch.manual_seed(MODEL_SEED)
    return MSFWSI(my_resnet.__dict__[arch], scale)


def build_case(man):
    """the product model of a golden case (architecture and hub-stub variant from its manifest)"""
    return build_product(man["arch"], residual_gain=man.get("stub_residual_gain", 1.0))


def case_batch(man, dtype=torch.float32):
    """the seeded input of a golden case: N(0,1) pixels ("normal") or the well-conditioned
    per-image patterns ("diverse", oracle.diverse_batch) -- regenerated from the manifest, never stored"""
    from oracle import msfwsi_oracle as orc

    return orc.make_batch(man.get("input_kind", "normal"), man["B"], man["size"], 16, man["data_seed"], dtype)


def load_golden(case):
    vec = dict(np.load(os.path.join(GOLDEN, case + ".npz")))
    with open(os.path.join(GOLDEN, case + ".json")) as f:
        man = json.load(f)
    return vec, man


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def flat_outputs(outs):
    """3x4x4 nested tuple -> {(group, kind, scale): tensor}"""
    res = {}
    for g, gname in zip(outs, ("context", "target", "fuser")):
        for kind, tup in zip(("p1", "p2", "z1", "z2"), g):
            for s, t in enumerate(tup):
                res[(gname, kind, s)] = t
    return res


def reference_loop_loss(outputs, weights=WEIGHTS):
    """the loss exactly as the reference loop writes it (tools/ssl_train.py:448-466), torch ops"""
    cos = torch.nn.CosineSimilarity(dim=1)
    loss = 0
    terms = []
    for grp in outputs:
        for i, (p1, p2, z1, z2) in enumerate(zip(*grp)):
            t = -(cos(p1.float(), z2.float()).mean() + cos(p2.float(), z1.float()).mean()) * 0.5
            terms.append(t.detach())
            loss = loss + t * weights[i]
    return loss, torch.stack(terms)


# --------------------------------------------------------------------------------------------------
# parity gates (north_star: 1e-3 relative fp32 on loss, embeddings, gradients, updated weights)
# --------------------------------------------------------------------------------------------------
FLOOR = 1e-3  # north_star tolerance


def spread_gate(rels, names, spreads, what, envelope=(), strict_count=False, tight_median=None, count_rule=True):
    """Per-tensor gate of SURVEY.md 8(d): rel-L2 against the fp64 oracle <= max(1e-3, 2 x the reference's OWN
    fp32<->fp64 spread of that tensor).

    rels     per-tensor rel-L2 of the product against the fp64 oracle
    spreads  samples of the reference's own per-tensor noise, same tensor order: the committed fixture
             (`spread_grad` / `spread_step`, real reference fp32 vs fp64, build container) and the oracle's fp32 vs
             fp64 run on THIS machine.  Per tensor the allowance uses the largest sample.
    envelope further samples of the same noise on other inputs (other fixture cases), used only for the size bound
             of rule 2.

    Rule 1 (per tensor): rel <= max(1e-3, 2 * spread).  The tensors that needed the spread allowance are printed.
    Rule 2 (outliers): the noise is a ReLU-gate flip process -- one pre-activation inside the forward rounding noise
             flips, and every gradient below it moves by 1e-3 .. 1e-2; each fp32 run of the REFERENCE shows these jumps
             on different tensors (compare the samples).  A product tensor beyond rule 1 is accepted only while the
             product shows no more than twice as many such tensors as the worst reference sample does (count of tensors
             above 1e-3), none of them further out than 2 x the largest spread any reference sample shows.  Nothing here is a free
             constant: every bound is read from the fixtures / the same-machine reference run.
    """
    rels = np.asarray(rels, dtype=np.float64)
    samples = [np.asarray(s, dtype=np.float64) for s in spreads]
    env = np.max(np.stack(samples), axis=0)
    allow = np.maximum(FLOOR, 2.0 * env)
    over = rels > allow
    needed = (rels > FLOOR) & ~over
    if needed.any():
        print(f"[{what}] {int(needed.sum())}/{len(rels)} tensors above 1e-3 but within 2x the reference's own "
              f"fp32<->fp64 spread:")
        for i in np.flatnonzero(needed)[:40]:
            print(f"    {names[i]}: rel {rels[i]:.2e}  (reference spread {env[i]:.2e})")
    n_ref = max(int((s > FLOOR).sum()) for s in samples + [np.asarray(e) for e in envelope])
    count_bound = 2 * n_ref
    if strict_count:
        # well-conditioned cases: most reference tensors sit far below 1e-3, so the count bound of rule 2 is below the
        # number of tensors and CAN trip (on the N(0,1) fixtures it could not: VERDICT r2, weak #2).  The count is of
        # TENSORS, the process is of EVENTS: one ReLU flip in a head (measured on the MI355X: target_projector.3) moves
        # every tensor of the encoder below it by 1.0-1.6e-3 -- 31 tensors from one event, where the reference's samples
        # of this case show 0 (fixture) and 9 (this box) -- so the bound is at least one encoder's worth of tensors (a
        # quarter of the model).  What a systematic error cannot pass is the median rule below: on these cases the
        # product's median must stay within 5x the reference's own (~2e-6), not merely below 1e-3.
        assert 2 * n_ref < len(rels), (what, "rule 2 cannot bite on this fixture", n_ref, len(rels))
        count_bound = max(count_bound, len(rels) // 4)
    if tight_median if tight_median is not None else strict_count:
        assert np.median(rels) <= 5.0 * max(float(np.median(env)), 1e-6), (what, "median (well-conditioned case)",
                                                                           float(np.median(rels)), float(np.median(env)))
    worst_ref = max(float(np.max(s)) for s in samples + [np.asarray(e) for e in envelope])
    if over.any():
        print(f"[{what}] {int(over.sum())} tensors beyond their own spread allowance (gate flips that the reference "
              f"samples show on OTHER tensors; the reference's worst sample has {n_ref} tensors above 1e-3, "
              f"largest spread {worst_ref:.2e}):")
        for i in np.flatnonzero(over)[:40]:
            print(f"    {names[i]}: rel {rels[i]:.2e}  (allowance {allow[i]:.2e})")
    print(f"[{what}] median {np.median(rels):.2e}  p90 {np.quantile(rels, 0.9):.2e}  max {rels.max():.2e} "
          f"({names[int(rels.argmax())]}); reference noise: median {np.median(env):.2e} max {env.max():.2e}")
    assert np.median(rels) <= max(FLOOR, 2.0 * float(np.median(env))), (what, "median", float(np.median(rels)))
    # (the same factor 2 as rule 1: the count of flip-hit tensors is itself a random number of the same process)
    # count_rule=False: a pure chain (an encoder trunk without heads), where ONE flip near the top moves EVERY tensor below
    # it -- the count of tensors then says where the flip sat, not how many there were (measured on the reference's own
    # fp32 runs of one trunk case: 13 tensors above 1e-3 with 16 torch threads, 153 of 159 with 128); the median and size
    # rules above / below still apply, and the caller adds a rule a flip cannot satisfy by luck (three-seed minimum)
    if count_rule:
        assert int(over.sum()) <= count_bound, (what, "tensors beyond their allowance", int(over.sum()), "bound", count_bound)
    if over.any():
        assert float(rels[over].max()) <= max(FLOOR, 2.0 * worst_ref), (what, names[int(rels.argmax())],
                                                                        float(rels[over].max()), worst_ref)
    return rels


def grad_rels(named_grads, ref):
    return np.array([rel(g, ref[n]) for n, g in named_grads])


def step_rels(named_params, ref_sd1):
    """per-tensor rel-L2 of the UPDATED weights (north_star: 'updated weights match ... within 1e-3')"""
    return np.array([rel(p, ref_sd1[n]) for n, p in named_params])


_ORACLE_CACHE = {}


def oracle_case(case):
    """One fp64 and one fp32 oracle step (forward, loss, backward, Adam) of a golden case on THIS machine, shared by
    the tests of a session: {B, size, batch, sd0, lr, loss64, terms64, outs64, grads64, sd64 (updated weights),
    loss32, sd32, names, box_grad, box_step (the oracle's own fp32<->fp64 spread per tensor here), vec, man}.
    The arithmetic runs in a background CPU worker when the session started some (tests/oracle_jobs.py, conftest.py:
    the GPU suite overlaps it with the GPU tests), otherwise inline."""
    if case in _ORACLE_CACHE:
        return _ORACLE_CACHE[case]
    import oracle_jobs

    vec, man = load_golden(case)
    out = dict(oracle_jobs.get("oracle_case", case))
    out["vec"], out["man"] = vec, man
    out["batch"] = case_batch(man)
    # pinned: this machine's fp64 oracle reproduces the real reference's golden loss terms
    assert torch.allclose(out["terms64"], torch.as_tensor(vec["terms"]), rtol=0, atol=1e-7)
    _ORACLE_CACHE[case] = out
    return out


def other_spreads(key, skip):
    """the same reference noise measured on the other fixture cases (size bound of spread_gate's rule 2)"""
    res = []
    for case in ("r18_b8_s64", "r18_b8_s224"):
        if case != skip:
            vec, _ = load_golden(case)
            if key in vec:
                res.append(vec[key])
    return res


ADAM_EPS = 1e-8  # torch.optim.Adam default, as tools/ssl_train.py uses it


def implied_gradient(w1, w0, g64, lr, tau=FLOOR):
    """The gradient that Adam's FIRST step says the run had: w1 - w0 = -lr * g / (|g| + eps), so with u = -(w1-w0)/lr,
    g = eps * u / (1 - |u|).  A saturated step (|u| >= 1 - tau: any |g| >= eps (1-tau)/tau gives it) carries no more
    than the sign: it is read as the reference gradient itself when that has the same sign and saturates too, else as
    the SMALLEST gradient of that sign that saturates -- so the result never overstates the agreement by more than the
    inversion can resolve, and never invents an error the update does not show."""
    u = -(w1 - w0) / lr
    a = u.abs()
    sat = a >= 1.0 - tau
    g_sat = ADAM_EPS * (1.0 - tau) / tau
    g = ADAM_EPS * u / (1.0 - a.clamp(max=1.0 - tau))
    same = sat & (torch.sign(u) == torch.sign(g64)) & (g64.abs() >= g_sat)
    return torch.where(same, g64, torch.where(sat, torch.sign(u) * g_sat, g))


def updated_weights_gate(named_params, sd0, sd64, grads64, lr, spreads, grad_spreads, what):
    """Updated weights after ONE Adam step against the fp64 oracle (north_star: 'updated weights ... within 1e-3').

    Per tensor: rel-L2(w1, w1_fp64) <= max(1e-3, 2 x the reference's own fp32<->fp64 spread of that tensor)
    (`spread_step` of the fixture / of this machine's oracle).  Adam's first step is -lr * g / (|g| + eps), i.e.
    -lr * sign(g) for all but vanishing gradients: an element whose gradient lies inside the gradient noise moves by
    +lr in one run and -lr in the other, and WHICH elements do so differs from run to run.  A tensor beyond its
    allowance is therefore examined element by element: where the update deviates from the reference's by more than
    1e-3 of a step, the gradient the update implies (`implied_gradient`) replaces the reference's, and the error of
    that implied gradient must fit inside the GRADIENT tolerance of the same tensor, max(1e-3, 2 x spread_grad) x
    ||g64|| -- the bound the gradient gate holds the product to.  No count and no constant of its own: the updated
    weights are accepted exactly when they are the Adam step of a gradient that is itself within tolerance."""
    names = [n for n, _ in named_params]
    env = np.max(np.stack([np.asarray(sp, dtype=np.float64) for sp in spreads]), axis=0)
    genv = np.max(np.stack([np.asarray(sp, dtype=np.float64) for sp in grad_spreads]), axis=0)
    rels, explained = [], []
    for (n, p), e, ge in zip(named_params, env, genv):
        w1 = torch.as_tensor(p).detach().double().cpu()
        ref1, w0 = sd64[n].double(), sd0[n].double()
        r = float((w1 - ref1).norm() / (ref1.norm() + 1e-300))
        rels.append(r)
        if r <= max(FLOOR, 2.0 * float(e)):
            continue
        g64 = grads64[n].double()
        dev = (w1 - ref1).abs() > FLOOR * lr
        gi = torch.where(dev, implied_gradient(w1, w0, g64, lr), g64)
        implied = float((gi - g64).norm()) / (float(g64.norm()) + 1e-300)
        allow_g = max(FLOOR, 2.0 * float(ge))
        assert implied <= allow_g, (what, n, "the update implies a gradient error beyond the gradient tolerance",
                                    int(dev.sum()), implied, allow_g)
        explained.append((n, r, int(dev.sum()), w1.numel(), implied, allow_g))
    rels = np.array(rels)
    if explained:
        print(f"[{what}] {len(explained)}/{len(names)} tensors beyond max(1e-3, 2 x spread), each the Adam step of a gradient "
              f"inside that tensor's gradient tolerance:")
        for n, r, k, tot, im, ag in explained[:40]:
            print(f"    {n}: rel {r:.2e}, {k}/{tot} elements deviate, implied gradient error {im:.1e} <= {ag:.1e}")
    print(f"[{what}] median {np.median(rels):.2e}  max {rels.max():.2e}; "
          f"{int((rels <= FLOOR).sum())}/{len(rels)} tensors within 1e-3")
    assert np.median(rels) <= max(FLOOR, 2.0 * float(np.median(env)))
    return rels


def gate_updated_weights(named_params, case, what):
    """updated weights of `case` after one Adam step against the fp64 oracle of this machine"""
    oc = oracle_case(case)
    assert [n for n, _ in named_params] == oc["names"]
    return updated_weights_gate(named_params, oc["sd0"], oc["sd64"], oc["grads64"], oc["lr"],
                                [oc["vec"]["spread_step"], oc["box_step"]],
                                [oc["vec"]["spread_grad"], oc["box_grad"]], what)


# --------------------------------------------------------------------------------------------------
# 16-bit gates: the yardstick is the REFERENCE under torch.autocast (fixture keys spread_*_bf16 / spread_*_fp16:
# distance of the reference's own 16-bit run from its fp64 run, tests/golden/make_golden.py)
# --------------------------------------------------------------------------------------------------
LOWP_FLOOR = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}  # two units in the last place of the storage type
LOWP_TAG = {torch.bfloat16: "bf16", torch.float16: "fp16"}


def lowp_gate(rels, names, ref_spread, floor, what, max_violations=0.05):
    """per tensor: rel-L2 of the product's 16-bit run against the fp64 oracle <= max(floor, 2 x the distance of the
    reference's own autocast run of that tensor from ITS fp64 run).  The 16-bit noise of one run is a sample, not a
    bound: up to `max_violations` of the tensors may exceed their own allowance, none by more than 2 x the LARGEST
    reference spread, and the median must stay within 1.5 x the reference's median."""
    rels = np.asarray(rels, dtype=np.float64)
    ref = np.asarray(ref_spread, dtype=np.float64)
    allow = np.maximum(floor, 2.0 * ref)
    over = rels > allow
    print(f"[{what}] product median {np.median(rels):.2e} max {rels.max():.2e} ({names[int(rels.argmax())]}); "
          f"reference-under-autocast median {np.median(ref):.2e} max {ref.max():.2e}; {int(over.sum())}/{len(rels)} "
          f"tensors beyond max({floor:.1e}, 2 x own reference spread)")
    for i in np.flatnonzero(over)[:20]:
        print(f"    {names[i]}: rel {rels[i]:.2e}  (allowance {allow[i]:.2e})")
    assert np.median(rels) <= max(floor, 1.5 * float(np.median(ref))), (what, "median", float(np.median(rels)))
    assert over.sum() <= max_violations * len(rels), (what, "tensors beyond their allowance", int(over.sum()))
    if over.any():
        assert float(rels[over].max()) <= max(floor, 2.0 * float(ref.max())), (what, names[int(rels.argmax())])
    return rels


import contextlib
import ctypes


@contextlib.contextmanager
def tuned(lib, settings):
    """msfwsi_set_tuning(key, value) for every (key, value) of `settings`, and on exit EXACTLY the values the switches had
    before (msfwsi_get_tuning): a fixture cannot leave the process-global switches in another state than it found them"""
    saved = {}
    try:
        for key, value in settings.items():
            old = ctypes.c_long()
            assert lib.msfwsi_get_tuning(int(key), ctypes.byref(old)) == 0, f"unknown tuning key {key}"
            saved[key] = old.value
            assert lib.msfwsi_set_tuning(int(key), int(value)) == 0
        yield
    finally:
        for key, value in saved.items():
            assert lib.msfwsi_set_tuning(int(key), int(value)) == 0
