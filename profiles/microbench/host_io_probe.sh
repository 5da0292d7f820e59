set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/host_io; mkdir -p $O; cd $R
g++ -O2 -pthread -o /tmp/host_io_probe profiles/microbench/host_io_probe.cpp || exit 1
for th in 8 16; do /tmp/host_io_probe /tmp/probe.txt ${1:-141000000} $th; done > $O/host_io_probe.txt 2>&1
cat $O/host_io_probe.txt; nproc; cat /sys/kernel/mm/transparent_hugepage/enabled; free -g | head -2
