cd $GRAFT_REPO_ROOT
timeout 2400 python tools/wg_hunt_loop.py --minutes 25 --arms wgsplit,wg --steps 3 --unfused-ln > gpurun_out/r06_hunt_unfused.log 2>&1
tail -6 gpurun_out/r06_hunt_unfused.log
