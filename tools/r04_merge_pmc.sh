#!/bin/bash
# development: merge the PMC directories of separate gpurun calls (tools/r04_prof.sh pmc1 / pmc2) into profiles/r04/pmc_counters.json
O=gpurun_out/r04p
python3 tools/pmc_collect.py profiles/r04/pmc_counters.json clips256=$O/pmc_clips256_fetch clips256=$O/pmc_clips256_write clips256=$O/pmc_clips256_sq1 clips256=$O/pmc_clips256_sq2 wave256=$O/pmc_wave256_fetch wave256=$O/pmc_wave256_write wave256=$O/pmc_wave256_sq1 wave256=$O/pmc_wave256_sq2 wave256f=$O/pmc_wave256f_fetch wave256f=$O/pmc_wave256f_write wave256f=$O/pmc_wave256f_sq1 wave256f=$O/pmc_wave256f_sq2 slide=$O/pmc_slide10_fetch slide=$O/pmc_slide10_write slide=$O/pmc_slide10_sq1 slide=$O/pmc_slide10_sq2 stream128=$O/pmc_stream128_fetch stream128=$O/pmc_stream128_write stream128=$O/pmc_stream128_sq1 stream128=$O/pmc_stream128_sq2 calib=$O/pmc_calib
cp $O/pmc_calib_memtime.jsonl profiles/r04/
