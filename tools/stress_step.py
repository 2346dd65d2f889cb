"""Randomised shape sweep of the whole training step (vlgae_amd/train_step.py, the reference's wiring): odd batch sizes, sentence lengths,
region counts, factor sets and widths.  Per case: (1) two independently built steps with one seed give the same bits (no race, no
uninitialised read that matters); (2) the float32 step with the fused feed-forwards equals the module-by-module formulation
(`fused_ff=False`) to 3e-2 relative L2 per gradient tensor (one flipped LeakyReLU branch in a small bias gradient is ~1e-2) -- dropout off, so both see the same function; (3) everything finite
in bf16.  Run on the GPU box: python tools/stress_step.py [seed] [cases]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import train_step
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 96]))      # (64, 96: enough token rows for the row-streaming input-gradient launches)
    L = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 40, 41, 47]))
    R = int(rng.choice([1, 2, 5, 16, 17, 36]))
    factors = [(), ("rel",), ("attr",), ("img",), ("rel", "attr", "img")][int(rng.integers(0, 5))]
    if "rel" in factors and R > 17:
        R = 17                                       # (R^2 relation columns: keep the sweep quick)
    kw = dict(d=int(rng.choice([64, 128])), h=int(rng.choice([64, 128, 256])), E=int(rng.choice([40, 104, 800])), n_vis=int(rng.choice([64, 256])),
              H=int(rng.choice([64, 256])), nb=int(rng.choice([0, 24, 150])), T=int(rng.choice([5, 45])), r=int(rng.choice([8, 16])),
              factors=factors, seed=int(rng.integers(1, 1000)))
    case = (it, B, L, R, kw)
    # (1) + (3): bf16 with dropout, built twice
    res = []
    for rep in range(2):
        step = train_step.build(B, L, R, dev, dtype=torch.bfloat16, **kw)
        loss, grads, _ = step()
        torch.cuda.synchronize()
        res.append((loss.detach().clone(), {k: v.detach().clone() for k, v in grads.items()}))
    assert torch.isfinite(res[0][0]).all() and all(torch.isfinite(v.float()).all() for v in res[0][1].values()), ("not finite", case)
    assert torch.equal(res[0][0], res[1][0]) and all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1]), ("not reproducible", case)
    # (2): float32, no dropout, fused against module-by-module
    off = dict(p_drop=0.0, p_enc=0.0, p_ff_drop=0.0, p_mid_drop=0.0)
    a = train_step.build(B, L, R, dev, dtype=torch.float32, **kw, **off)
    b = train_step.build(B, L, R, dev, dtype=torch.float32, fused_ff=False, **kw, **off)
    la, ga, _ = a()
    lb, gb, _ = b()
    assert abs(float(la) - float(lb)) <= 2e-5 * max(1.0, abs(float(lb))), ("loss", case, float(la), float(lb))
    if torch.equal(a.last["heads"], b.last["heads"]):      # (a near-tie may flip the Viterbi tree between two fp32 formulations: compare on equal trees)
        gmax = max(float(v.abs().max()) for v in gb.values())
        for k in ga:
            # (two float32 formulations: a LeakyReLU pre-activation within rounding of zero takes different branches -- slope 0.01 against 1 --
            #  so single elements may differ by more than rounding; an indexing error would be O(1) in the norm)
            ref = gb[k].double()
            e = float((ga[k].double() - ref).norm() / max(float(ref.norm()), 1e-3 * gmax * ref.numel() ** 0.5))
            assert e <= 3e-2, ("gradient", case, k, e)
            worst = max(worst, e)
    print("case %d ok: B=%d L=%d R=%d factors=%s d=%d h=%d E=%d H=%d nb=%d T=%d r=%d" % (it, B, L, R, "+".join(factors) or "-", kw["d"], kw["h"], kw["E"],
                                                                                       kw["H"], kw["nb"], kw["T"], kw["r"]), flush=True)
print("stress ok; worst fused-vs-module gradient difference %.2e (relative L2)" % worst)
