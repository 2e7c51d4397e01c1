cd $GRAFT_REPO_ROOT
python -m pytest tests/test_recurrent.py -m gpu -x -q 2>&1 | tail -25
python scripts/run_config.py config4 --iterations 4 2>&1 | grep iteration
