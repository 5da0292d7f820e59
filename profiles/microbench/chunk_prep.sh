# One dosplitalign chunk of the end-to-end shape (100 k fusions): the host's set-up stage by stage (DEFUSE_TIMING).   gpurun -- bash profiles/microbench/chunk_prep.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e2e; mkdir -p $O; D=/tmp/e2e_prep
python3 $R/profiles/microbench/e2e_scale.py --fragments 2000000 --support 10 30 --no-fused --parallel 1 --out $D > $O/prep_chain.json 2> $O/prep_chain.err || { tail -20 $O/prep_chain.err; exit 1; }
C="$R/bin/dosplitalign -f $D/ref.fa -e $D/exons.txt -u 450 -s 45 -n 150 -x 150 -r $D/clusters.sc.regions -i $D/improper.0.sam -1 $D/reads.0.1.fastq -2 $D/reads.0.2.fastq -a $D/prep.split"
for k in 1 2 3; do DEFUSE_TIMING=1 $C 2> $O/chunk_prep_$k.txt; done
grep -E "\[tasks\]|regions|tasks \(|fasta index \+|main\(\)|ready" $O/chunk_prep_3.txt
