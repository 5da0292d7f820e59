# kernel_stats.sh for an ablation build: $1 = name of build_var/lib_<name>.so
export DEFUSE_DSA_LIB=$GRAFT_REPO_ROOT/build_var/lib_$1.so
bash $GRAFT_REPO_ROOT/profiles/microbench/kernel_stats.sh | grep -E "k_replay|k_emit"
