set -u
bash tools/gpu.sh prof r06_prof --no-search --no-extra
bash tools/gpu.sh pmc r06_pmc_hbm hbm --no-search
bash tools/gpu.sh pmc r06_pmc_sq sq --no-search
