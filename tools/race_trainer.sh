cd $GRAFT_REPO_ROOT
MSFWSI_TUNING=15=1 timeout -k 10 300 python -u tools/race_check.py resnet18 16 64 bf16 6 trainer-poison || exit 1
MSFWSI_TUNING=15=1 timeout -k 10 300 python -u tools/race_check.py resnet18 16 64 fp32 4 trainer-poison || exit 1
MSFWSI_TUNING=15=1 timeout -k 10 300 python -u tools/race_check.py resnet18 16 64 bf16 6 trainer-poison || exit 1
MSFWSI_TUNING=15=1 timeout -k 10 600 python -u tools/race_check.py resnet50 8 64 bf16 3 trainer-poison || exit 1
