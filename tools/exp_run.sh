for g in 16 32 64; do for dm in 16 32 64; do echo "fp6 games $g dmax $dm"; OMOK_FC0_DMAX=$dm python tools/play_plies.py 15 $g 800 16 3 3 2>/dev/null | tail -1 | cut -c60-; done; done
for g in 8 16 32 64; do for dm in 32 64; do echo "mixed games $g dmax $dm"; OMOK_FC0_DMAX=$dm python tools/play_plies.py 15 $g 800 16 3 5 2>/dev/null | tail -1 | cut -c60-; done; done
