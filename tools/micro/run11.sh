# panel width of the tile order (SVL_PLANES_PANEL): time of the ViT shapes + PMC fetch of FFN-1
cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for pw in 0 3 4 6; do echo "== panel $pw"; for args in "32800 3072 768 30 4" "32800 768 3072 30 2" "32800 2304 768 30 0" "32800 768 768 30 2"; do echo -n "$args: "; SVL_PLANES_PANEL=$pw timeout 120 $B $args | tail -1; done; done
