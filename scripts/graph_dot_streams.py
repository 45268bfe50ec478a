import re,sys,collections
txt=open(sys.argv[1]).read()
nodes={}
for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*?label="(\d+)\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)',txt):
    nid=int(m.group(1)); nodes[nid]=dict(name=m.group(3),stream=int(m.group(4)),sig=m.group(5))
edges=[(int(a),int(b)) for a,b in re.findall(r'"graph_1_node_(\d+)"\s*->\s*"graph_1_node_(\d+)"',txt)]
print(len(nodes),'nodes',len(edges),'edges')
def short(n):
    n=re.sub(r'_ZN2at6native\d*|_ZN12_GLOBAL__N_1\d+|_ZN7sdeconv\d+|_GLOBAL__N_1\d*','',n)
    return n[:38]
pred=collections.defaultdict(list); succ=collections.defaultdict(list)
for a,b in edges: pred[b].append(a); succ[a].append(b)
cnt=collections.Counter(n['stream'] for n in nodes.values()); print('nodes per stream',dict(cnt))
if len(sys.argv)>3:
    lo,hi=int(sys.argv[2]),int(sys.argv[3])
    for i in range(lo,hi):
        if i in nodes:
            print(i,'s%d'%nodes[i]['stream'],short(nodes[i]['name']),'<-',pred[i],'->',succ[i])
if len(sys.argv)>2 and sys.argv[2]=='runs':
    ids=sorted(nodes)
    start=ids[0]; cur=nodes[start]['stream']
    for a,b in zip(ids,ids[1:]+[None]):
        if b is None or nodes[b]['stream']!=cur:
            print(f"nodes {start:3d}-{a:3d} s{cur}  {short(nodes[start]['name'])} ... {short(nodes[a]['name'])}   first<-{pred[start]}  last->{succ[a]}")
            if b is not None: start=b; cur=nodes[b]['stream']
