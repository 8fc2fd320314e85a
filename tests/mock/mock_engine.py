"""GPU-free stand-in for the `omok_ai_amd` package surface that bench.py uses (OMOK_BENCH_ENGINE=mock.mock_engine): lets a
CPU test run `python bench.py --gpus 2` end to end (self-launch, rank environment, sharding by game_offset, the gloo /
RCCL-shaped gather of replay records, the result line) without a GPU.  It measures nothing: every number is synthetic."""
import json
import os
import types

import numpy as np
import torch

import omok_ai_amd as _real

dist = _real.dist
weights = _real.weights
binding = types.SimpleNamespace(NET_F16X3=0, NET_F32=1, NET_F16X3_ROWS=2, NET_F16X3_FP6=3, NET_F16X3_F16=4, NET_F16X3_MIXED=5, FC0_FORMATS={-1: "f32", 0: "fp6", 1: "f16", 2: "mixed"})


class Engine:
    def __init__(self, board_size=15, games=1, max_nodes=0, max_tables=0, max_batch_k=16, device=0, net_mode=0, seed=0, game_offset=0):
        self.n, self.hw, self.games, self.game_offset, self.seed = board_size, board_size * board_size, games, game_offset, seed
        self.episodes = 0
        out = os.environ.get("OMOK_MOCK_DIR")
        if out:
            with open(os.path.join(out, f"rank{os.environ.get('RANK', '0')}.json"), "w") as f:
                json.dump({"game_offset": game_offset, "games": games, "device": device, "world": os.environ.get("WORLD_SIZE")}, f)

    def load_random_weights(self, seed=0):
        pass

    def set_profiling(self, on=True):
        pass

    def reset_stats(self):
        self.episodes = 0

    def stats(self):
        g, e = float(self.games), float(self.episodes)
        return {"sims": 800 * 10 * g * e, "evals": 790 * 10 * g * e, "ply_games": 10 * g * e, "finished": g * e, "ms_tree": 1.0, "ms_trunk": 5.0,
                "ms_fc0": 3.0, "ms_tail": 0.5, "ms_ply": 0.2, "fc0_launches": 50 * e, "fc0_rows": 790 * 10 * g * e, "tree_bytes": 1e6,
                "round_launches": 50 * e, "ms_round": 0.8, "peak_nodes": 100.0, "peak_tables": 10.0}

    def close(self):
        pass


class SelfPlay:
    def __init__(self, eng):
        self.eng = eng

    def reset(self):
        pass

    def run(self, *a):
        self.eng.episodes += 1
        return self.eng.stats()

    def game_info(self):
        g = self.eng.games
        return np.zeros(g, np.uint8), np.full(g, 2, np.uint8), np.full(g, 10, np.int32)

    def replay_record_bytes(self):
        return (self.eng.hw + 1 + 3) // 4 * 4 + 4 * self.eng.hw + 4

    def replay_pack_into(self, ptr, cap):
        """rank r ships (r + 1) * games records whose first 8 bytes carry the global game id: uneven counts on purpose"""
        return -1  # (bench.py's mock path fills the buffer itself through pack_tensor)

    def pack_tensor(self, buf, rec):
        rank = int(os.environ.get("RANK", "0"))
        cnt = (rank + 1) * self.eng.games
        view = buf[: cnt * rec].view(cnt, rec)
        view.zero_()
        ids = torch.arange(cnt, dtype=torch.int64) % self.eng.games + self.eng.game_offset
        view[:, :8] = ids.view(-1, 1).contiguous().view(torch.uint8).view(cnt, 8)
        return cnt
