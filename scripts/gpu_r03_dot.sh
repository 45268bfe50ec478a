# the instantiated step graphs (kernel nodes, edges, the runtime's stream assignment) as Graphviz files: DEBUG_HIP_GRAPH_DOT_PRINT writes them into the cwd
mkdir -p gpurun_out/prof
D=$GRAFT_REPO_ROOT/gpurun_out/prof
dump() { tag=$1; shift; rm -rf /tmp/dotdir && mkdir -p /tmp/dotdir && cd /tmp/dotdir && DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout -k 10 200 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 5 --warmup 2 "$@" > $D/r03u_$tag.json 2> $D/r03u_$tag.err; ls /tmp/dotdir; for f in /tmp/dotdir/graph_*; do cp $f $D/r03u_${tag}_$(basename $f).dot; done; cd $GRAFT_REPO_ROOT; }
dump mono18_multi --workload mono_r18 --opt photo_multi=1
dump sup50 --workload sup_r50
dump mono50 --workload mono_r50
ls -la $D | grep r03u
