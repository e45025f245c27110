"""How the generator's reference framing (host/scene_gen.cpp, flags bit 3) was fitted to /root/reference/readme/ps5_b5_s128.png:
the image's 8x8 pixel blocks that are exactly black (38.8 %), their share per band of rows and of columns (16 bands each:
REF_ROWS / REF_COLS below, measured with PIL from the PNG), against the CPU oracle's primary-hit mask of the generated scene
(480x270, 2x2 blocks) under a camera at azimuth az, distance r from the axis, height h, looking at (tx, ty, 0), over a ground
scaled by gscale.  Random search over the six numbers:
    python tools/framing_fit.py search          ->  az 0.82, r 8.6, h 6.47, ty 1.87, gscale 0.539, tx 1.0: 0.387 empty
    python tools/framing_fit.py check           ->  the same figures for the generator's flag itself
TEST INFRASTRUCTURE side only (uses the oracle); nothing here runs in the product."""
import sys, time, numpy as np, ctypes as C
sys.path.insert(0,'.')
import __graft_entry__ as e
pta=e.load_package(); oracle=e.load_oracle()
def look_at(P,T):
    P=np.array(P,float); T=np.array(T,float)
    f=(T-P)/np.linalg.norm(T-P); r=np.cross(f,[0,1,0]); r/=np.linalg.norm(r); u=np.cross(r,f)
    M=np.zeros(16,np.float32)
    cols=[r,u,-f,P]
    for k in range(4):
        M[4*k:4*k+3]=cols[k]; M[4*k+3]=1.0 if k==3 else 0.0
    return M
def evaluate(az,r,h,ty,gscale,tx=0.0,W=480,H=270,tris=20000):
    sc=pta.HostScene.generate_ps5(tris,0,0)
    d=sc.desc.contents
    n_ground = None
    # ground = model 0
    m0=d.models[0]
    tri=np.ctypeslib.as_array(d.triangles,(int(d.n_triangles)*24,)).reshape(-1,3,8)
    g=tri[m0.tri_first:m0.tri_first+m0.tri_count]
    g[:,:,0]*=gscale; g[:,:,2]*=gscale
    P=(r*np.sin(az)+tx, h, r*np.cos(az))
    M=look_at(P,(tx,ty,0))
    for i in range(16): d.camera.transform[i]=float(M[i])
    osc=oracle.OracleScene(sc.desc, oracle.PTO_BVH)
    pl=osc.debug_render(W,H)
    if not pl: return 1.0,None
    hit=(pl['ior'].reshape(H,W,3).max(axis=2)>0)   # ior plane: 1/3*255 where hit
    b=hit.reshape(H//2,2,W//2,2).max(axis=(1,3))
    return 1-b.mean(), b
if __name__=='__main__' and len(sys.argv)==1:
    for args in [(0.0,9.0,2.4,1.9,1.0),(0.7,9.0,5.0,1.9,1.0),(0.7,9.0,5.0,1.9,0.5),(0.7,8.0,4.5,1.9,0.5),(0.75,8.5,5.5,2.0,0.45)]:
        t=time.time(); f,b=evaluate(*args); 
        rows=[round(float(1-x.mean()),2) for x in np.array_split(b,16,axis=0)]
        print(args, 'empty frac', round(f,3), 'rows', rows, round(time.time()-t,1),'s')

def profiles(b):
    rows=np.array([1-x.mean() for x in np.array_split(b,16,axis=0)]); cols=np.array([1-x.mean() for x in np.array_split(b,16,axis=1)])
    return rows, cols
REF_ROWS=np.array([0.98, 0.88, 0.83, 0.78, 0.77, 0.68, 0.48, 0.28, 0.1, 0.01, 0.0, 0.0, 0.01, 0.03, 0.02, 0.0])
REF_COLS=np.array([0.54, 0.5, 0.46, 0.42, 0.36, 0.26, 0.16, 0.11, 0.12, 0.34, 0.39, 0.43, 0.47, 0.51, 0.56, 0.6])
def cost(p):
    f,b=evaluate(*p)
    if b is None: return 10
    r,c=profiles(b)
    return ((r-REF_ROWS)**2).sum()+((c-REF_COLS)**2).sum()+10*(f-0.3877)**2
def search(seed=0, iters=400):
    rng=np.random.default_rng(seed)
    best=None
    lo=np.array([0.3,5.0,2.5,1.2,0.3,-1.0]); hi=np.array([1.2,12.0,8.0,2.8,1.0,1.0])
    cur=np.array([0.75,8.0,4.5,1.9,0.5,0.0]); cc=cost(cur); best=(cc,cur.copy())
    step=(hi-lo)*0.15
    for i in range(iters):
        cand=np.clip(cur+rng.normal(size=6)*step,lo,hi)
        c=cost(cand)
        if c<cc: cur,cc=cand,c
        if c<best[0]: best=(c,cand.copy()); 
        if i%50==49: step*=0.7; print(i,best, flush=True)
    return best
if len(sys.argv)>1 and sys.argv[1]=='search':
    b=search()
    f,bb=evaluate(*b[1]); r,c=profiles(bb)
    print('best',b,'frac',f); print('rows',np.round(r,2)); print('cols',np.round(c,2))

if len(sys.argv)>1 and sys.argv[1]=='check':
    sc=pta.HostScene.generate_ps5(20000,0,8)
    osc=oracle.OracleScene(sc.desc, oracle.PTO_BVH)
    W,H=480,270
    pl=osc.debug_render(W,H)
    hit=(pl['ior'].reshape(H,W,3).max(axis=2)>0)
    b=hit.reshape(H//2,2,W//2,2).max(axis=(1,3))
    r,c=profiles(b)
    print('generator flags 8: empty 8x8-equivalent blocks', round(float(1-b.mean()),4), '(reference image: 0.3877)')
    print('rows', np.round(r,2)); print('ref ', REF_ROWS); print('cols', np.round(c,2)); print('ref ', REF_COLS)
    print('rms deviation', round(float(np.sqrt((((r-REF_ROWS)**2).sum()+((c-REF_COLS)**2).sum())/32)),3))
