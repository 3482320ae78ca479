"""Shared helpers for the parity tests: oracle/product model pairs on identical weights."""
import torch

from visitron_amd.synth import deterministic_state_dict


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def model_pair(oracle_cls, product_cls, cfg, seed=0, device="cuda:0", weight_std=0.05):
    ref = oracle_cls(cfg).eval()
    ref.load_state_dict(deterministic_state_dict(ref, seed=seed, weight_std=weight_std))
    prod = product_cls(cfg).eval()
    prod.load_state_dict(ref.state_dict())
    if hasattr(prod, "tie_weights"):
        prod.tie_weights()
    return ref, prod.to(device)


def maxabs(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())
