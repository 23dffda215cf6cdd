cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
