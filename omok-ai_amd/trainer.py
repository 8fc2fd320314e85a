"""One-GPU (or one-process-per-GPU) mirror of Trainer::train (src/trainer.rs:69-386) over the HIP engine.

Per iteration, like the reference: clear the replay memory (:77-78), play `episode_count` self-play games (:81-205, on
the engine), back-fill z and augment (:207-324, on the device), cap the memory at `replay_memory_size` (:326-328), run
`parameter_update_count` training steps on `parameter_update_batch_size` transitions (:329-357), save the model
(`saves/<model_name>`, :375, ModelIO format).  Plots and the periodic games against the naive player (:371-400) are not
part of this mirror.  Parameters and defaults: src/config.rs:83-110 (episode_count 50, evaluate_count 600, ...).
Multi-GPU: every rank plays its own `episode_count` games (global ids rank*episode_count + g) and trains data-parallel
(gradients averaged per step), so all ranks hold identical weights after every iteration.
"""
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import api, dist, weights
from . import train as T


@dataclass
class Parameters:  # src/config.rs:83-110
    model_name: str = "alpha-zero"
    replay_memory_size: int = 600_000
    episode_count: int = 50
    evaluate_count: int = 600
    evaluate_batch_size: int = 16
    epsilon: float = 0.25
    alpha: float = 0.03
    temperature: float = 1.0
    temperature_threshold: int = 30
    parameter_update_count: int = 600
    parameter_update_batch_size: int = 128


class Trainer:
    def __init__(self, params=None, board_size=15, seed=0, save_dir="saves", max_nodes=None, max_tables=None, precision_rows=256, precision_search_rounds=True):
        self.p = params or Parameters()
        self.n = board_size
        self.rank, self.local_rank, self.world = dist.shard_info()
        self.device = f"cuda:{self.local_rank}"
        self.save_dir = save_dir
        self.precision_rows = precision_rows  # independent check of the net outputs after every weight update (0 = off)
        self.precision_search_rounds = precision_search_rounds  # ... and of the search rounds' own path (sibling base + difference rows)
        self.last_precision = None
        sims = -(-self.p.evaluate_count // self.p.evaluate_batch_size) * self.p.evaluate_batch_size
        max_nodes = max_nodes or min(16384, 4 * sims + 1024)
        self.engine = api.Engine(board_size=board_size, games=self.p.episode_count, max_nodes=max_nodes,
                                 max_tables=max_tables or max(256, max_nodes // 4), max_batch_k=self.p.evaluate_batch_size,
                                 device=self.local_rank, seed=seed, game_offset=dist.game_offset(self.rank, self.p.episode_count))
        path = os.path.join(save_dir, self.p.model_name)
        if os.path.exists(path):  # Trainer::new -> this.load(model_name) (:63-65, :628-636)
            self.engine.load(path)
            tensors = self._engine_tensors()
        else:
            tensors = weights.init_random(board_size, seed=seed)
            self.engine.load_weights(tensors)
        self.phase = T.TrainPhase(board_size, tensors, self.device)  # the optimizer state lives across iterations like the session's
        self.selfplay = api.SelfPlay(self.engine)
        self.iteration = 0
        it_path = path + ".iteration"  # (not in the reference, whose thread_rng is fresh on every start): a resumed run must
        if os.path.exists(path) and os.path.exists(it_path):  # not replay the RNG streams of the iterations it has already played;
            self.iteration = int(open(it_path).read().strip() or 0)  # a counter without its checkpoint is stale and ignored

    def _engine_tensors(self):
        tmp = os.path.join(self.save_dir, f".{self.p.model_name}.rank{self.rank}.tmp")
        self.engine.save(tmp)
        from . import model_file
        tensors = model_file.load(tmp)[1]
        os.remove(tmp)
        return tensors

    def train(self, iteration_count, log=print):
        p = self.p
        rec = self.selfplay.replay_record_bytes()
        for _ in range(iteration_count):
            self.selfplay.set_episode(self.iteration)  # RNG stream of this iteration (key = seed + iteration * golden ratio)
            self.iteration += 1
            self.engine.reset_stats()  # (per-iteration counters in the log line)
            self.selfplay.reset()  # fresh agents; the engine's replay buffer is cleared with them (:77-93)
            stats = self.selfplay.run(p.evaluate_count, p.evaluate_batch_size, p.epsilon, p.alpha, p.temperature,
                                      p.temperature_threshold, 0)
            _, _, plies = self.selfplay.game_info()
            total = 6 * int(plies.sum())
            buf = torch.empty(max(total, 1) * rec, dtype=torch.uint8, device=self.device)
            got = self.selfplay.replay_augment_into(buf.data_ptr(), total)
            records = buf[: got * rec].reshape(got, rec)
            if got > p.replay_memory_size:  # pop_front until the memory fits (:326-328)
                records = records[got - p.replay_memory_size:]
            v_loss, p_loss, loss = self.phase.run(records, p.parameter_update_count, p.parameter_update_batch_size,
                                                  seed=self.iteration * 7919 + self.rank)
            self.phase.push_to(self.engine)
            # new weights -> omok_net_commit -> the engine re-measured fc0's operand format on its probe set; the probe's
            # figures are kept for the log, and the outputs are checked independently below (a probe is a measurement, not a proof)
            st = self.engine.stats()
            fmt = api.B.FC0_FORMATS[int(st["fc0_format"])]  # the format THIS engine runs (its probe saw plain rows and, if its rounds are large enough, a sibling round)
            self.last_precision = {"fc0_format": fmt, "probe_rows": int(st["probe_rows"]), "probe_outside": int(st["probe_outside"]),
                                   "probe_fp6": (st["probe_dp_fp6"], st["probe_dv_fp6"]), "probe_f16": (st["probe_dp_f16"], st["probe_dv_f16"])}
            if self.precision_rows > 0 and fmt != "f32":  # independent check on rows of this iteration's replay buffer (spread over the buffer); on by default
                # The check engines are FORCED into the training engine's format: left to their own probes they would validate whatever a 64-game engine chooses
                # (never mixed).  A failure of the check (e.g. no memory for its engines) is logged, never fatal: the training state above is already consistent.
                try:
                    from . import precision
                    forced = precision.FORCED_MODE[fmt]
                    idx = torch.linspace(0, records.shape[0] - 1, min(self.precision_rows, records.shape[0]), device=records.device).long()
                    x, _, _ = T.decode_records(records[idx], self.n)
                    chk = precision.measure(self.phase.net.tensors(), self.n, x.reshape(x.shape[0], -1).cpu().numpy(), device=self.local_rank,
                                            batch_k=p.evaluate_batch_size, net_mode=forced)
                    self.last_precision["check"] = chk
                    if chk["fc0_format"] != fmt:
                        log(f"[iter={self.iteration}] WARNING: the precision check ran in format {chk['fc0_format']}, the engine runs {fmt}")
                    if not chk["within_contract"]:
                        log(f"[iter={self.iteration}] WARNING: net outputs differ from the fp32 kernels by |dp| {chk['max_dp']:.2e} |dv| {chk['max_dv']:.2e} "
                            f"(contract 1e-3) in format {chk['fc0_format']}")
                    if self.precision_search_rounds:  # the path the search rounds take (base row + 7x7-window difference rows), on rounds of the new net
                        games = precision.difference_path_games(self.n, p.evaluate_batch_size)  # (enough rows per round for the difference path at either board size)
                        sr = precision.measure_search_rounds(self.phase.net.tensors(), self.n, games=games, batch_k=p.evaluate_batch_size, rounds=2, plies=1,
                                                             device=self.local_rank, seed=self.iteration, net_mode=forced)
                        self.last_precision["search_rounds"] = sr
                        if sr["difference_path_rounds"] == 0:
                            log(f"[iter={self.iteration}] note: no round of the search-round check took the difference path ({sr['rows']} rows checked on the copy path)")
                        if not sr["within_contract"]:
                            log(f"[iter={self.iteration}] WARNING: search-round outputs differ from the fp32 kernels by |dp| {sr['max_dp']:.2e} |dv| {sr['max_dv']:.2e} "
                                f"(contract 1e-3) on {sr['rows']} rows")
                except Exception as ex:  # noqa: BLE001
                    self.last_precision["check_error"] = repr(ex)
                    log(f"[iter={self.iteration}] WARNING: the precision check did not run: {ex!r}")
            if self.rank == 0:  # Trainer::save (:605-626)
                os.makedirs(self.save_dir, exist_ok=True)
                final = os.path.join(self.save_dir, p.model_name)  # counter first, then the weights, each by rename: a crash in between
                with open(final + ".iteration.tmp", "w") as f:     # leaves the OLD weights with the NEW counter (an RNG stream is skipped,
                    f.write(str(self.iteration))                   # never replayed)
                os.replace(final + ".iteration.tmp", final + ".iteration")
                self.engine.save(final + ".tmp")
                os.replace(final + ".tmp", final)
            log(f"[iter={self.iteration}] games={int(stats['finished'])} transitions={got} loss={loss:.4f} "
                f"[v_loss={v_loss:.4f}, p_loss={p_loss:.4f}]")
        return v_loss, p_loss, loss

    def close(self):
        self.engine.close()
