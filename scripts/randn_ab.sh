for m in 1 0 1 0; do echo DIRECT=$m; ITTS_RANDN_DIRECT=$m python3 scripts/run_section.py world 256 16000 25.0 False False 2>&1 | grep -i "ms\|rtf" | head -12; done
