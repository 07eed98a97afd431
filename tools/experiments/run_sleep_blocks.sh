bash tools/experiments/run_timeline.sh | grep -E "averaged|k_colour_count|k_sleep_apply|k_setup_slots|k_colour_inherit|k_cache_build"
