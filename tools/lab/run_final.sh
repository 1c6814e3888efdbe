cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/lab/final_profiles.sh 2>&1 | tail -12
