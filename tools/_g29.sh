bash tools/refresh_round.sh r4 2>&1 | tail -2
