# development aid (round 4): parity and timing of the arm and the point robot with the development library
bash tests/tools/dev_r04_arm.sh || exit 1
bash tests/tools/dev_r04_stamps.sh
