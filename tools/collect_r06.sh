#!/bin/bash
# Round-6 evidence on the GPU box (run from the repo root): tools/collect_r06.sh gpurun_out/r06 [part...]   parts: tests bench kstats pmc c3 misc driver (default: all)
out=${1:-gpurun_out/r06}; shift; parts=${@:-tests bench kstats pmc c3 misc driver}; mkdir -p $out; R=$PWD
has() { [[ " $parts " == *" $1 "* ]]; }
keep() { cp gpurun_out/bench_full.json $out/$1 2>/dev/null; }
if has tests; then
  echo "[tests]"; python -m pytest tests -q -m gpu -s > $out/gputests_full.log 2>&1; tail -2 $out/gputests_full.log
fi
if has bench; then
  echo "[bench c2, 3 steps]"; python3 bench.py --steps 3 --warmup 1 > $out/bench_c2_full_episode.json 2> $out/bench_c2.err; tail -c 300 $out/bench_c2_full_episode.json; echo; keep bench_c2_full_episode_verbose.json
fi
if has kstats; then
  echo "[kernel stats c2]"; tools/kstats.sh $out c2 --max-plies 4 | head -24
  cp $out/ks_c2.log $out/bench_c2_first4plies_under_rocprofv3.log 2>/dev/null
fi
if has pmc; then
  echo "[pmc c2]"; tools/collect_pmc.sh $out 15 | tail -14
fi
if has c3; then
  echo "[bench c3]"; python3 bench.py --board 9 --games 16384 --sims 200 --batch-k 8 --steps 2 --warmup 1 --cpu-seconds 20 > $out/bench_c3_9x9_16384games.json 2> $out/bench_c3.err; tail -c 300 $out/bench_c3_9x9_16384games.json; echo; keep bench_c3_verbose.json
  tools/kstats.sh $out c3 --board 9 --games 16384 --sims 200 --batch-k 8 --max-plies 4 | head -16
fi
if has misc; then
  echo "[bench c1]"; python3 bench.py --games 1 --sims 100 --batch-k 16 --cpu-seconds 0 > $out/bench_c1_single_game.json 2>/dev/null; tail -c 200 $out/bench_c1_single_game.json; echo
  echo "[rehearsal]"; OMOK_BENCH_BACKEND=gloo python bench.py --gpus 2 --games 256 --sims 64 --max-plies 6 --gather --cpu-seconds 0 > $out/rehearsal_2ranks_one_gpu_gloo.json 2> $out/rehearsal.err; tail -c 300 $out/rehearsal_2ranks_one_gpu_gloo.json; echo
  echo "[soak]"; (python tools/soak_determinism.py 1024 512 2 15 16 auto; python tools/soak_determinism.py 2048 200 2 9 8 auto) > $out/soak_determinism.log 2>&1; tail -3 $out/soak_determinism.log
  echo "[ply times]"; python tools/ply_times.py > $out/ply_times_by_live_games.txt 2>&1; tail -12 $out/ply_times_by_live_games.txt
fi
if has driver; then
  echo "[driver-style bench: --gpus 1 --steps 20 --warmup 5]"; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c2_20steps_driver_style.json 2> $out/bench_driver.err; tail -1 $out/bench_c2_20steps_driver_style.json | head -c 400; echo; keep bench_c2_20steps_driver_style_verbose.json
fi
