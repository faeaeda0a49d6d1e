"""Diagnostic (CPU): rows, widths and coverage of the small triangles (raster classes 0 and 1) of a workload, per tile entry --
what k_raster's row list holds.  usage: python tools/raster_stats.py [c3]"""
import numpy as np, sys
sys.path.insert(0,'.')
from bibim_renderer_amd import configs
from oracle import bbo, scenes
cfg = configs.CONFIGS[sys.argv[1] if len(sys.argv)>1 else "c3"]
ball = scenes.load_shaderball_vertices()
inst = scenes.ball_instances(cfg.grid)
vu = scenes.view_uniforms(cfg.cam_pos, cfg.cam_yaw, cfg.cam_pitch, cfg.width, cfg.height, 1, cfg.fov, cfg.near, cfg.far)
P = np.array(vu["proj"], np.float64).reshape(4,4).T   # column-major M[col][row] -> matrix
V = np.array(vu["view"], np.float64).reshape(4,4).T
pos = np.concatenate([ball["pos"].astype(np.float64), np.ones((len(ball),1))],1)
W,H = cfg.width, cfg.height
tot = {0:[0,0,0,0,0.0,[]],1:[0,0,0,0,0.0,[]]}
for i in range(len(inst)):
    M = np.array(inst[i]["model"], np.float64).reshape(4,4).T
    c = pos @ (P@V@M).T
    w = c[:,3]
    ok = w > 0.1
    x = (c[:,0]/w+1)*W/2; y = (c[:,1]/w+1)*H/2
    X = np.rint(x*256).reshape(-1,3); Y = np.rint(y*256).reshape(-1,3); okt = ok.reshape(-1,3).all(1)
    S = (X[:,1]-X[:,0])*(Y[:,2]-Y[:,0]) - (X[:,2]-X[:,0])*(Y[:,1]-Y[:,0])
    front = (S>0)&okt
    X=X[front];Y=Y[front];S=S[front]
    minX=X.min(1);maxX=X.max(1);minY=Y.min(1);maxY=Y.max(1)
    px0=np.maximum(np.ceil((minX-128)/256),0);px1=np.minimum(np.floor((maxX-128)/256),W-1)
    py0=np.maximum(np.ceil((minY-128)/256),0);py1=np.minimum(np.floor((maxY-128)/256),H-1)
    vis=(px0<=px1)&(py0<=py1)
    px0,px1,py0,py1,S,minX,maxX,minY,maxY=[a[vis] for a in (px0,px1,py0,py1,S,minX,maxX,minY,maxY)]
    ext=np.maximum(maxX-minX,maxY-minY); area=(px1-px0+1)*(py1-py0+1)
    cls=np.where((ext<=32*256)&(area<=64),0,np.where(ext<=64*256,1,2))
    for c_ in (0,1):
        m=cls==c_
        for a0,a1,b0,b1,s in zip(px0[m],px1[m],py0[m],py1[m],S[m]):
            # per tile pieces
            for ty in range(int(b0)//32,int(b1)//32+1):
                r0=max(b0,ty*32); r1=min(b1,ty*32+31)
                for tx in range(int(a0)//32,int(a1)//32+1):
                    c0=max(a0,tx*32); c1=min(a1,tx*32+31)
                    t=tot[c_]; t[0]+=1; t[1]+=r1-r0+1; t[2]+=(r1-r0+1)*(c1-c0+1); t[5].append(c1-c0+1)
            tot[c_][3]+=1; tot[c_][4]+=s/2/65536
for c_ in (0,1):
    t=tot[c_]; w=np.array(t[5])
    print(f"class {c_}: triangles {t[3]} tile-entries {t[0]} rows {t[1]} ({t[1]/max(t[0],1):.1f}/entry) tested px {t[2]} ({t[2]/max(t[1],1):.1f}/row) triangle area px {t[4]:.0f} coverage {t[4]/max(t[2],1):.2f}  width pct 50/90/99: {np.percentile(w,[50,90,99])}")
