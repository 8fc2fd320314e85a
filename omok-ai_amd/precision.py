"""Precision of the product net path (split-operand MFMA) against the OMOK_NET_F32 kernels of the same library on the same GPU, for a
given set of weights and input rows.  The contract (BASELINE.json north_star) is 1e-3 on the outputs of AgentModel::evaluate_pv
(alpha-zero/src/agent_model.rs:116-134): p after softmax, v after tanh.

`omok_net_commit` measures fc0's operand formats against the fp32 kernels -- 2048 synthetic positions through the plain-row path and, where
the engine's rounds are large enough for it, one synthetic round of sibling runs through the difference path -- and keeps the fastest of
fp6 (block-scaled fp6 correction terms, products good to ~2^-15), mixed (f16 correction terms on full rows, fp6 on the difference rows:
+2..3 %) and f16 (~2^-22, +20 %) whose worst |dp|, |dv| stay within 3e-4 and whose logits stay within 5e-4 (`Engine.stats()`:
fc0_format, probe_*).  The probe is a measurement, not a proof: held-out positions have exceeded its figures by up to 1.7x (what the
limits are sized for).  The functions below are the independent check on rows of the caller's choosing: tests,
bench.py and `Trainer` (on by default) use them.

At board_size 15 the SEARCH ROUNDS take a different path through the first two stages of the net (sibling requests = one base row
+ 7x7-window difference rows) than `omok_evaluate_pv`; `measure_search_rounds` checks that path on the request rows of
real rounds."""
import numpy as np

from . import api
from . import binding as B


FORCED_MODE = {"fp6": B.NET_F16X3_FP6, "f16": B.NET_F16X3_F16, "mixed": B.NET_F16X3_MIXED}  # the net mode that forces a format name (Engine.stats()["fc0_format"])


def measure(tensors, n, inputs, device=0, batch_k=16, net_mode=B.NET_F16X3):
    """dict(max_dp, max_dv, max_dlogit, max_dvpre, logit_abs_max, rows) for float32 `inputs` [R, 3 * n * n] (encoder.rs layout).
    `net_mode`: the product mode to check -- a caller that wants the verdict on the format ITS engine runs passes FORCED_MODE[that format]
    (a 64-game check engine has no difference-path rounds, so its own probe can only choose fp6 or f16)."""
    x = np.ascontiguousarray(inputs, dtype=np.float32).reshape(len(inputs), -1)
    out = {}
    res = []
    for mode in (net_mode, B.NET_F32):
        eng = api.Engine(board_size=n, games=64, max_nodes=8, max_tables=4, max_batch_k=batch_k, device=device, net_mode=mode)
        eng.load_weights(tensors)
        if mode != B.NET_F32:
            fmt = B.FC0_FORMATS[int(eng.stats()["fc0_format"])]
        p, v = eng.evaluate_pv(x)
        lg, vp = eng.evaluate_logits(x)
        eng.close()
        res.append((p.reshape(len(x), -1), v.reshape(-1), lg, vp))
    (p, v, lg, vp), (p32, v32, lg32, vp32) = res
    out["rows"] = int(len(x))
    out["max_dp"] = float(np.abs(p - p32).max())
    out["max_dv"] = float(np.abs(v - v32).max())
    out["max_dlogit"] = float(np.abs(lg - lg32).max())
    out["max_dvpre"] = float(np.abs(vp - vp32).max())
    out["logit_abs_max"] = float(np.abs(lg32).max())
    out["logit_std"] = float(lg32.std())
    out["within_contract"] = bool(out["max_dp"] < 1e-3 and out["max_dv"] < 1e-3)
    out["fc0_format"] = fmt
    return out


def difference_path_games(n, batch_k):
    """games an engine needs for its search rounds to take the difference path in every operand format (rounds of >= 3072 rows at board_size 15, >= 1024 at 9)"""
    return max(64, -(-(3072 if n == 15 else 1024) // batch_k))


def measure_search_rounds(tensors, n, games=64, batch_k=16, rounds=6, plies=3, device=0, seed=1, net_mode=B.NET_F16X3):
    """The same comparison for the outputs of search rounds: plays `plies` plies of `rounds` rounds on `games` trees with the
    product engine and compares every round's p / v with the fp32 kernels' evaluation of the same request rows.  (Only rounds of
    >= 3072 rows (board_size 15; 1024 at 9) take the difference path in every format: games * batch_k must reach that for the check to cover it --
    `difference_path_games`; the result says how many of the rounds did: `difference_path_rounds`.)"""
    eng = api.Engine(board_size=n, games=games, max_nodes=max(1024, 2 * rounds * batch_k), max_tables=256, max_batch_k=batch_k, device=device,
                     seed=seed, net_mode=net_mode)
    eng.load_weights(tensors)
    ref = api.Engine(board_size=n, games=games, max_nodes=8, max_tables=4, max_batch_k=batch_k, device=device, net_mode=B.NET_F32)
    ref.load_weights(tensors)
    sp = api.SelfPlay(eng)
    sp.reset()
    out = {"rows": 0, "max_dp": 0.0, "max_dv": 0.0, "max_dlogit": 0.0, "max_dvpre": 0.0, "fc0_format": B.FC0_FORMATS[int(eng.stats()["fc0_format"])]}
    for _ in range(plies):
        if sp.alive_count == 0:
            break
        for rnd in range(rounds):
            nreq = sp.round_generate(rnd, batch_k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            p, v = np.array(p).reshape(nreq, -1), np.array(v).reshape(-1)
            lg, vp = sp.round_logits()  # (the quantities in front of softmax / tanh, on the path the round took)
            sp.round_scatter()
            if nreq:
                p32, v32 = ref.evaluate_pv(x)
                lg32, vp32 = ref.evaluate_logits(x)
                out["max_dp"] = max(out["max_dp"], float(np.abs(p - p32.reshape(nreq, -1)).max()))
                out["max_dv"] = max(out["max_dv"], float(np.abs(v - v32.reshape(-1)).max()))
                out["max_dlogit"] = max(out["max_dlogit"], float(np.abs(lg - lg32.reshape(nreq, -1)).max()))
                out["max_dvpre"] = max(out["max_dvpre"], float(np.abs(vp - vp32.reshape(-1)).max()))
                out["rows"] += int(nreq)
        sp.sample_actions(1.0, 30)
        sp.advance()
    out["difference_path_rounds"] = int(eng.stats()["children2_launches"])  # (0: every round took the copy / plain path -- the check covered none of the difference path)
    eng.close()
    ref.close()
    out["within_contract"] = bool(out["max_dp"] < 1e-3 and out["max_dv"] < 1e-3)
    out["logits_within_1e-3"] = bool(out["max_dlogit"] < 1e-3 and out["max_dvpre"] < 1e-3)
    return out


def search_round_rows(tensors, n, games, batch_k=16, warm_plies=3, warm_sims=64, rounds=2, device=0, seed=1, net_mode=B.NET_F16X3):
    """Request rows of real search rounds AND what the product engine computed for them on the path those rounds took: plays `warm_plies` plies, then
    `rounds` step-wise rounds; returns dict(x [R, 3 n n] encoder.rs rows, p, v, logits, vpre, fc0_format, difference_path_rounds, rows_per_round).  The caller
    compares with a reference of its own.  With games * batch_k >= 3072 (board 15) the rounds take the difference path."""
    eng = api.Engine(board_size=n, games=games, max_nodes=max(1024, 4 * (warm_sims + rounds * batch_k)), max_tables=256, max_batch_k=batch_k, device=device,
                     seed=seed, net_mode=net_mode)
    eng.load_weights(tensors)
    sp = api.SelfPlay(eng)
    sp.reset()
    sp.run(warm_sims, batch_k, 0.25, 0.03, 1.0, 30, warm_plies)
    eng.reset_stats()
    xs, ps, vs, ls, vps, per_round = [], [], [], [], [], []
    for rnd in range(rounds):
        nreq = sp.round_generate(rnd, batch_k, 0.25, 0.03)
        x = sp.round_inputs().copy()
        p, v = sp.round_eval()
        lg, vp = sp.round_logits()
        sp.round_scatter()
        per_round.append(int(nreq))
        if nreq:
            xs.append(x.reshape(nreq, -1)); ps.append(np.array(p).reshape(nreq, -1)); vs.append(np.array(v).reshape(-1))
            ls.append(np.array(lg).reshape(nreq, -1)); vps.append(np.array(vp).reshape(-1))
    st = eng.stats()
    eng.close()
    return {"x": np.concatenate(xs), "p": np.concatenate(ps), "v": np.concatenate(vs), "logits": np.concatenate(ls), "vpre": np.concatenate(vps),
            "fc0_format": B.FC0_FORMATS[int(st["fc0_format"])], "difference_path_rounds": int(st["children2_launches"]), "rows_per_round": per_round}
