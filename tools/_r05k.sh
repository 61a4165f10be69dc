bash tools/prof.sh r05k tests default
bash tools/prof.sh r05 stats3 stats1 stats_ss pmc
