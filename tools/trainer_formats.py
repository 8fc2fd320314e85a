"""Evidence for DESIGN 3.4: a few iterations of the Trainer mirror (self-play on the engine, replay post-processing, Adadelta steps, weights
pushed back through omok_net_commit) with the commit probe's figures and the operand format it chose after every weight update, plus an
independent check of the net outputs against the fp32 kernels on rows of the iteration's replay buffer.
usage: python tools/trainer_formats.py [board] [iterations] [episode_count] [evaluate_count] [update_count]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omok_ai_amd import trainer as TR

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
its = int(sys.argv[2]) if len(sys.argv) > 2 else 4
par = TR.Parameters(episode_count=int(sys.argv[3]) if len(sys.argv) > 3 else 64, evaluate_count=int(sys.argv[4]) if len(sys.argv) > 4 else 64,
                    evaluate_batch_size=8 if n == 9 else 16, parameter_update_count=int(sys.argv[5]) if len(sys.argv) > 5 else 300)
with tempfile.TemporaryDirectory() as d:
    t = TR.Trainer(par, board_size=n, seed=0, save_dir=d, precision_rows=512)
    for i in range(its):
        t.train(1, log=lambda s: print(s, flush=True))
        lp = t.last_precision
        ck = lp.get("check", {})
        print(f"  iteration {i + 1}: format {lp['fc0_format']} (probe verdict {lp['probe_outside']}); probe fp6 |dp| {lp['probe_fp6'][0]:.2e} |dv| {lp['probe_fp6'][1]:.2e}, f16 {lp['probe_f16'][0]:.2e} / {lp['probe_f16'][1]:.2e}; "
              f"replay rows vs fp32 kernels in format {ck.get('fc0_format')}: |dp| {ck.get('max_dp', float('nan')):.2e} |dv| {ck.get('max_dv', float('nan')):.2e} "
              f"(|logit| max {ck.get('logit_abs_max', float('nan')):.1f}) within contract: {ck.get('within_contract')}", flush=True)
    t.close()
