cd $GRAFT_REPO_ROOT
( for V in "MM_PAIR_GROUP=64" "MM_PAIR_GROUP=64"; do echo "$V: $(env $V timeout 300 python tools/heaviest_chain_tick.py 10000000 2 2>&1 | grep -v amdgpu.ids | tail -1)"; done ) > gpurun_out/r06e_heaviest.txt 2>&1
MM_PAIR_DEBUG=1 timeout 200 python tools/heaviest_chain_tick.py 10000000 2 2>&1 | grep -E "g0 fast|tile1 cycles" | tail -2 > gpurun_out/r06e_timers.txt
