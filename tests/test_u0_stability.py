"""Decision H1 under test: upstream `solve()` with no argument starts from a RANDOM vector
([REF roman/align/object_registration.py:27] calls `clipper.solve()`; SURVEY.md B5: uniform[0,1) from std::random_device),
this tree from the all-ones vector.  SURVEY.md §7 "hard part 2" asks for the stability of the solution across seeds to be
REPORTED: for K random starts per problem, how often is the selected association set the one the all-ones start finds?

CPU part (oracle only, small cases): runs everywhere, prints the fractions, asserts that the planted problems are stable.
GPU part: K = 16 random starts per problem at BASELINE config 1, config 2 and 32 problems of config 3 — the GPU must equal
the oracle for EVERY start (explicit u0 is part of the C ABI), and the fraction of starts that reproduce the all-ones
result is printed (and returned in `tests/_u0_report.json` when ROMAN_U0_REPORT is set, for DESIGN.md)."""
import json
import os

import numpy as np
import pytest

from conftest import registration_for
from roman_amd import synth
from roman_amd.align import batch as rb

K_SEEDS = 16


def _starts(nA, k, seed):
    rng = np.random.default_rng(seed)
    return [rng.random(nA) for _ in range(k)]           # uniform [0,1), as upstream's randvec()


def _as_set(assoc):
    return frozenset(map(tuple, np.asarray(assoc).reshape(-1, 2).tolist()))


CPU_CASES = [("cfg1", "clipper", {}, 30, 30, 0, 1000),
             ("semgrav60", "semanticgrav", {"semantics_dim": 64}, 60, 50, 64, 12),
             ("roman50", "roman", {"semantics_dim": 32}, 50, 50, 32, 13),
             ("gravity40", "gravity", {}, 40, 40, 0, 11)]


@pytest.mark.parametrize("case", CPU_CASES, ids=[c[0] for c in CPU_CASES])
def test_oracle_solution_is_stable_across_random_starts(orc, case):
    _, method, kw, n, m, d, seed = case
    reg = registration_for(method, **kw)
    P = reg._abi_params()
    pr = synth.make_pair(n, m, d, seed, tilt_deg=1.0 if P.gravity_guided else 0.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    base = _as_set(orc.register(P, D1, D2)["assoc"])
    same, recall = 0, []
    truth = _as_set(pr.inliers)
    for u0 in _starts(n * m, K_SEEDS, 77 + seed):
        got = _as_set(orc.register(P, D1, D2, u0=u0)["assoc"])
        same += int(got == base)
        recall.append(len(got & truth) / max(len(truth), 1))
    print(f"[u0 stability] {case[0]}: {same}/{K_SEEDS} random starts select the all-ones set; planted recall min {min(recall):.2f}")
    assert min(recall) >= 0.8                       # every start finds the planted clique
    assert same >= K_SEEDS // 2                      # and mostly the very same set


@pytest.mark.gpu
def test_gpu_equals_oracle_for_every_random_start_and_reports_stability(ctx, orc):
    report = {}
    near_ties = []

    def run(tag, reg, pairs, seed0):
        reg.set_context(ctx)
        P = reg._abi_params()
        batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
        nA = (batch.n1.astype(np.int64) * batch.n2).tolist()
        base = rb.run_batch(reg, batch)
        same = np.zeros(len(pairs), dtype=int)
        for k in range(K_SEEDS):
            u0s = [_starts(nA[b], 1, seed0 + 1000 * k + b)[0] for b in range(len(pairs))]
            res = rb.run_batch(reg, batch, u0=np.concatenate(u0s))
            # the oracle on all problems of this start side by side (one thread per problem)
            many = orc.register_many(P, batch.feats, batch.off1, batch.n1, batch.off2, batch.n2, batch.kmax(), u0=np.concatenate(u0s))
            for b in range(len(pairs)):
                if not np.array_equal(res.assoc[b], many[b]):
                    D1 = batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]]; D2 = batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]]
                    o = orc.register(P, D1, D2, u0=u0s[b])          # (the full solution: u is needed to judge an order difference)
                    # The SET must be the oracle's.  The ORDER (descending u) of two entries may differ only where their u
                    # values are closer than the iteration's own convergence tolerance (tol_u = 1e-8; the solve stops with u
                    # known to ~1e-8, and the oracle's own two arithmetic modes then order such a pair differently too —
                    # seen on config 3, seed 3025, start 1: gap 1.4e-8).
                    assert _as_set(res.assoc[b]) == _as_set(o["assoc"]), f"{tag}: problem {b}, start {k}: GPU selects another set than the oracle"
                    uo = o["u"][o["assoc"][:, 0] * int(batch.n2[b]) + o["assoc"][:, 1]]
                    rows = np.nonzero(np.any(res.assoc[b] != o["assoc"], axis=1))[0]
                    assert all(abs(uo[r] - uo[min(r + 1, len(uo) - 1)]) < 1e-6 or abs(uo[r] - uo[max(r - 1, 0)]) < 1e-6 for r in rows), \
                        f"{tag}: problem {b}, start {k}: order differs away from a near-tie"
                    near_ties.append((tag, b, k))
                same[b] += int(_as_set(res.assoc[b]) == _as_set(base.assoc[b]))
        frac = float(same.sum()) / (K_SEEDS * len(pairs))
        report[tag] = {"problems": len(pairs), "starts_per_problem": K_SEEDS, "fraction_equal_to_all_ones": frac,
                       "problems_stable_under_every_start": int((same == K_SEEDS).sum())}
        print(f"[u0 stability] {tag}: {same.sum()}/{K_SEEDS * len(pairs)} random starts select the all-ones set "
              f"({(same == K_SEEDS).sum()}/{len(pairs)} problems stable under every start)")

    run("config1", registration_for("clipper"), [synth.make_pair(30, 30, 0, 1000)], 11)
    run("config2", registration_for("semanticgrav", semantics_dim=512), [synth.make_pair(200, 200, 512, 2000)], 12)
    run("config3_first32", registration_for("semanticgrav", semantics_dim=512),
        [synth.make_pair(200, 200, 512, 3000 + k) for k in range(32)], 13)
    report["order_only_differences_at_near_ties"] = len(near_ties)
    print(f"[u0 stability] order-only differences at near-ties (|du| < 1e-6): {near_ties}")
    out = os.environ.get("ROMAN_U0_REPORT")
    if out:
        with open(out, "w") as fh:
            json.dump(report, fh, indent=1)
    assert all(v["fraction_equal_to_all_ones"] >= 0.5 for v in report.values() if isinstance(v, dict))
