bash tests/tools/dev_r04_arm.sh || exit 1
bash tests/tools/dev_r04_stamps.sh
