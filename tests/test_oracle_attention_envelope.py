"""How reproducible is the REFERENCE's own cascade under its bf16 attention?  (CPU only; test infrastructure like everything that imports oracle/.)

ppmstereo.py:550 calls flash_attn.flash_attn_func on bf16 operands.  FlashAttention-2 (absent here: third-party, restated from its published
algorithm) keeps the scores and the softmax statistics in fp32 but rounds the unnormalised probabilities P~ = exp(S - m) to
the operand type (bf16) for the P~ V product, key block by key block under a running maximum m; its key-block size depends on head dimension and
architecture.  The committed fixtures were generated with the simplest restatement (fp32 P, one block).  This test evaluates the oracle's cascade
at iters=20 (10/10/20 iterations, 40 predictions: configs 3-5's iteration count) with three equally valid restatements of that product -- P~ in
bf16 under the global maximum, and the blockwise form with 64- and 128-key blocks -- and measures how far the 40 predictions move.

Finding (asserted below, table printed with pytest -s, committed as profiles/r05_attention_rounding_envelope.txt): any two of the four differ by
0.7-1.2e-3 px mean at predictions 9-29 and 1.1-1.5e-3 px at prediction 39, while two fp32 evaluation orders with the SAME attention rounding differ
by 3.5e-4 (tests/test_oracle_golden.py).  The recurrence amplifies the bf16 rounding of P~: at 40 predictions the reference is reproducible to
~1.2e-3 px across FlashAttention block orders, so north_star's 1e-3 px budget is met by the GPU path at iters=10 (4.7e-4) and is below the
reference's own reproducibility at iters=20, where the GPU path (1.32e-3 px at prediction 39, tests/test_gpu_zz_full_configs.py) sits inside this
envelope."""
import numpy as np
import torch

from oracle import ppm_oracle as O
from test_oracle_golden import W, it10_cascade_inputs


def _bf16p_global(Q, K, V, scale):
    Qb, Kb, Vb = (t.to(torch.bfloat16).float() for t in (Q, K, V))
    S = (Qb @ Kb.t()) * scale
    Pt = torch.exp(S - S.max(dim=-1, keepdim=True).values)
    return ((Pt.to(torch.bfloat16).float() @ Vb) / Pt.sum(-1, keepdim=True)).to(torch.bfloat16).float()


def _bf16p_blocks(bk):
    def fa(Q, K, V, scale):           # FlashAttention-2 forward, Algorithm 1: running maximum m, running sum l, rescaled fp32 accumulator
        Qb, Kb, Vb = (t.to(torch.bfloat16).float() for t in (Q, K, V))
        n = Q.shape[0]
        m, l, acc = torch.full((n, 1), -float("inf")), torch.zeros(n, 1), torch.zeros(n, V.shape[1])
        for s in range(0, K.shape[0], bk):
            S = (Qb @ Kb[s:s + bk].t()) * scale
            mn = torch.maximum(m, S.max(-1, keepdim=True).values)
            a, Pt = torch.exp(m - mn), torch.exp(S - mn)
            l = l * a + Pt.sum(-1, keepdim=True)
            acc = acc * a + Pt.to(torch.bfloat16).float() @ Vb[s:s + bk]
            m = mn
        return (acc / l).to(torch.bfloat16).float()
    return fa


def test_reference_cascade_reproducibility_across_flash_attention_block_orders(monkeypatch):
    T, feats = it10_cascade_inputs()
    variants = (("fp32 P (fixtures)", O.flash_attn_math), ("bf16 P, global max", _bf16p_global), ("bf16 P, 64-key blocks", _bf16p_blocks(64)),
                ("bf16 P, 128-key blocks", _bf16p_blocks(128)))
    res = {}
    for name, fn in variants:
        monkeypatch.setattr(O, "flash_attn_math", fn)
        preds, uncs = [], []
        O.cascade(W, feats, 20, T, preds, uncs)
        assert len(preds) == 40
        res[name] = torch.stack(preds).numpy()
    names = [n for n, _ in variants]
    shown = (0, 4, 9, 19, 29, 39)
    print("\nmean |difference| in px of the oracle's 40 cascade predictions (T=5, 64x256, iters=20) between restatements of the P~ V rounding")
    print(f"{'':>24s} vs {'':<24s}" + "".join(f"  pred {i:2d}" for i in shown))
    pair = {}
    for a in range(len(names)):
        for b in range(a + 1, len(names)):
            d = np.abs(res[names[a]] - res[names[b]]).reshape(40, -1).mean(1)
            pair[(a, b)] = d
            print(f"{names[a]:>24s} vs {names[b]:<24s}" + "".join(f" {d[i]:.2e}" for i in shown))
    # (1) the rounding of P~ alone moves the last prediction past the 1e-3 px budget, for every pair of restatements ...
    for k, d in pair.items():
        assert d[39] > 8e-4, (k, d[39])
        assert d[39] < 3e-3 and d[:30].mean() < 1.5e-3, (k, d[39], d[:30].mean())      # ... but stays a rounding-level effect (no divergence)
    # (2) ... including two blockwise FlashAttention-2 orders against each other: the reference's own reproducibility at 40 predictions
    assert pair[(2, 3)][39] > 8e-4
    # (3) at iters=10's depth (20 predictions: 5/5/10) the effect is inside the budget region the GPU tests assert (< 1e-3 px mean)
    assert all(d[0] < 5e-4 for d in pair.values())
