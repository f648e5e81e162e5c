cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; mkdir -p $O
B="python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample"
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $O/a -o a -- $B > $O/a.log 2>&1; echo "rc_a $?" >> $O/rc.txt
timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQC_TC_INST_REQ --kernel-trace --output-format csv -d $O/b -o b -- $B > $O/b.log 2>&1; echo "rc_b $?" >> $O/rc.txt
ls -R $O | head -30
