import os, sys, numpy as np, tempfile
sys.path.insert(0, os.getcwd())
os.environ["SKDER_AMD_DEBUG_FASTA"]="1"
import skder_amd
rng=np.random.RandomState(1); alpha=np.frombuffer(b"ACGT",np.uint8)
g=bytes(alpha[rng.randint(0,4,30000)])
d=tempfile.mkdtemp()
w=lambda b: b"\n".join(b[i:i+80] for i in range(0,len(b),80))+b"\n"
cases={"a.fa": b"\n>rec0 some description\n"+w(g[:1129])+b">rec1 x\n"+w(g[1129:]),
       "b.fa": b">rec0 some description\n"+w(g[:1129])+b">rec1 x\n"+w(g[1129:]),
       "c.fa": b"\n\n\n>rec0 some description\n"+w(g[:600])+b">rec1 x\n"+w(g[600:]),
       "d.fa": b"\n>s\nACGT\n>rec0 some description\n"+w(g[:600])+b">rec1 x\n"+w(g[600:])}
for k,v in cases.items(): open(os.path.join(d,k),"wb").write(v)
l=os.path.join(d,"l.txt"); open(l,"w").write("".join(os.path.join(d,k)+"\n" for k in sorted(cases)))
skder_amd.runSkaniTriangle(l, os.path.join(d,"o.tsv"), "-s 80", 0.0, "greedy", False, None)
print(open(os.path.join(d,"o.tsv")).read())
