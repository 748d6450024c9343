# round 6: the whole GPU suite under knob settings that push every launch through other schedules (none can change a pixel)
O=gpurun_out/r6_suite_variants; mkdir -p $O
export AMD_LOG_LEVEL=1
run() { name=$1; shift; env "$@" python -m pytest tests -m gpu -q --capture=sys -x > $O/$name.log 2>&1; echo "$name ($*): $(tail -1 $O/$name.log)" | tee -a $O/summary.txt; }
run chunks2 RPT_UNIT_ROUNDS=64 RPT_UNIT_MIN_SPP=2
run sdf_room_1 RPT_SDF_SHADE_ROOM=1 RPT_SDF_MARCH_MIN_LANES=3
run sdf_room_64 RPT_SDF_SHADE_ROOM=64 RPT_SDF_MARCH_MIN_LANES=16
run thresholds RPT_SHADE_THRESHOLD=20 RPT_FINISH_THRESHOLD=60
