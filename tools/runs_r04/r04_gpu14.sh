mkdir -p gpurun_out/r04
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=30 ) > gpurun_out/r04/r04_gputest_full.log 2>&1
tail -45 gpurun_out/r04/r04_gputest_full.log
