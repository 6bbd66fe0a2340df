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


def install_hub_stub():
    """offline torch.hub stand-in: an un-pretrained net of the same arch under HUB_SEED, built with the
    product's own factories; the caller's RNG stream is untouched (same protocol as make_golden.py)."""
    from msf_wsi_amd.models import resnet as my_resnet

    def fake(url, progress=True, **kw):
        arch = [k for k, v in my_resnet.model_urls.items() if v == url][0]
        state = torch.random.get_rng_state()
        torch.manual_seed(HUB_SEED)
        sd = my_resnet.__dict__[arch](pretrained=False).state_dict()
        torch.random.set_rng_state(state)
        return sd

    torch.hub.load_state_dict_from_url = fake


def build_product(arch="resnet18", scale=4):
    from msf_wsi_amd.models import resnet as my_resnet
    from msf_wsi_amd.models.backbone import MSFWSI

    install_hub_stub()
    torch.manual_seed(MODEL_SEED)
    return MSFWSI(my_resnet.__dict__[arch], scale)


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
