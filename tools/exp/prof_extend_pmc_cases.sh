for c in 1,1024,0 1,4096,0 4,512,2048; do TAG=$(echo $c | tr , _) CASE=$c bash tools/exp/prof_extend_pmc.sh || exit 1; done
