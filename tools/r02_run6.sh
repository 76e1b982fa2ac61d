cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/r02_gputest_full.log 2>&1; echo "exit $?" >> $O/r02_gputest_full.log
tail -12 $O/r02_gputest_full.log
bash tools/profile_round.sh r02 2>&1 | tail -50
