cd $GRAFT_REPO_ROOT
for k in 0 2 4 6 8 12 16 24 0; do MISO_TUNE=$((k*256)) python3 tools/train_ab.py 2>/dev/null | tail -1 | sed "s/^/k=$k /"; done
