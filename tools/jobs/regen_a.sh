# round artifacts, part A: bench line, kernel stats, HBM traffic, SQ counters
set -e
R=$(pwd)
bash tools/profile_round.sh r02
bash tools/pmc_sq.sh r02
