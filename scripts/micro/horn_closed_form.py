import numpy as np
rng=np.random.default_rng(0)
def rod(r):
    th=np.linalg.norm(r); k=r/th; K=np.array([[0,-k[2],k[1]],[k[2],0,-k[0]],[-k[1],k[0],0]])
    return np.eye(3)+np.sin(th)*K+(1-np.cos(th))*K@K
def buildN(X,M):
    Cs=X.mean(0); Ce=M.mean(0)
    s=np.zeros((3,3))
    for a in range(3):
        for j in range(3):
            s[a,j]=(X[:,a]*M[:,j]).sum()/3-Ce[j]*Cs[a]
    s_=s.reshape(-1)
    Q=np.zeros((4,4))
    Q[0,0]=s_[0]+s_[4]+s_[8]; Q[1,1]=s_[0]-s_[4]-s_[8]; Q[2,2]=s_[4]-s_[8]-s_[0]; Q[3,3]=s_[8]-s_[0]-s_[4]
    Q[1,0]=Q[0,1]=s_[5]-s_[7]; Q[2,0]=Q[0,2]=s_[6]-s_[2]; Q[3,0]=Q[0,3]=s_[1]-s_[3]
    Q[2,1]=Q[1,2]=s_[3]+s_[1]; Q[3,1]=Q[1,3]=s_[6]+s_[2]; Q[3,2]=Q[2,3]=s_[7]+s_[5]
    return s,Q,Cs,Ce
def Rq(q):
    q0,q1,q2,q3=q
    return np.array([[q0*q0+q1*q1-q2*q2-q3*q3,2*(q1*q2-q0*q3),2*(q1*q3+q0*q2)],[2*(q1*q2+q0*q3),q0*q0+q2*q2-q1*q1-q3*q3,2*(q2*q3-q0*q1)],[2*(q1*q3-q0*q2),2*(q2*q3+q0*q1),q0*q0+q3*q3-q1*q1-q2*q2]])
def closed(X,M):
    s,N,Cs,Ce=buildN(X,M)
    cof=np.array([[s[(i+1)%3,(j+1)%3]*s[(i+2)%3,(j+2)%3]-s[(i+1)%3,(j+2)%3]*s[(i+2)%3,(j+1)%3] for j in range(3)] for i in range(3)])
    lam=np.sqrt((s*s).sum()+2*np.sqrt((cof*cof).sum()))
    a=N-lam*np.eye(4)
    s0=a[0,0]*a[1,1]-a[1,0]*a[0,1]; s1=a[0,0]*a[1,2]-a[1,0]*a[0,2]; s2=a[0,0]*a[1,3]-a[1,0]*a[0,3]
    s3=a[0,1]*a[1,2]-a[1,1]*a[0,2]; s4=a[0,1]*a[1,3]-a[1,1]*a[0,3]; s5=a[0,2]*a[1,3]-a[1,2]*a[0,3]
    c5=a[2,2]*a[3,3]-a[3,2]*a[2,3]; c4=a[2,1]*a[3,3]-a[3,1]*a[2,3]; c3=a[2,1]*a[3,2]-a[3,1]*a[2,2]
    c2=a[2,0]*a[3,3]-a[3,0]*a[2,3]; c1=a[2,0]*a[3,2]-a[3,0]*a[2,2]; c0=a[2,0]*a[3,1]-a[3,0]*a[2,1]
    b=np.zeros((4,4))
    b[0,0]= a[1,1]*c5-a[1,2]*c4+a[1,3]*c3; b[0,1]=-a[0,1]*c5+a[0,2]*c4-a[0,3]*c3; b[0,2]= a[3,1]*s5-a[3,2]*s4+a[3,3]*s3; b[0,3]=-a[2,1]*s5+a[2,2]*s4-a[2,3]*s3
    b[1,0]=-a[1,0]*c5+a[1,2]*c2-a[1,3]*c1; b[1,1]= a[0,0]*c5-a[0,2]*c2+a[0,3]*c1; b[1,2]=-a[3,0]*s5+a[3,2]*s2-a[3,3]*s1; b[1,3]= a[2,0]*s5-a[2,2]*s2+a[2,3]*s1
    b[2,0]= a[1,0]*c4-a[1,1]*c2+a[1,3]*c0; b[2,1]=-a[0,0]*c4+a[0,1]*c2-a[0,3]*c0; b[2,2]= a[3,0]*s4-a[3,1]*s2+a[3,3]*s0; b[2,3]=-a[2,0]*s4+a[2,1]*s2-a[2,3]*s0
    b[3,0]=-a[1,0]*c3+a[1,1]*c1-a[1,2]*c0; b[3,1]= a[0,0]*c3-a[0,1]*c1+a[0,2]*c0; b[3,2]=-a[3,0]*s3+a[3,1]*s1-a[3,2]*s0; b[3,3]= a[2,0]*s3-a[2,1]*s1+a[2,2]*s0
    j=np.argmax(np.abs(np.diag(b)))
    q=b[:,j]/np.linalg.norm(b[:,j])
    R=Rq(q); T=Ce-R@Cs
    return R,T,lam
def eig(X,M):
    s,N,Cs,Ce=buildN(X,M)
    w,v=np.linalg.eigh(N)
    q=v[:,-1]; R=Rq(q); return R,Ce-R@Cs,w
worst=0; worst_inc=0
for t in range(20000):
    X=rng.uniform(-2000,2000,(3,3))
    R0=rod(rng.normal(size=3)); T0=rng.uniform(-1000,1000,3)+np.array([0,0,2500])
    M=X@R0.T+T0
    inc = 10**rng.uniform(-14,-1)
    M=M*(1+inc*rng.normal(size=(3,1)))   # incongruent: scale each ray length
    R1,T1,lam=closed(X,M); R2,T2,w=eig(X,M)
    d=max(np.abs(R1-R2).max(), np.abs(T1-T2).max()/2500)
    gap=(w[-1]-w[-2])/w[-1]
    if d>worst: worst=d; print(t,'d',d,'inc',inc,'gap',gap,'lam err',abs(lam-w[-1])/w[-1])
print('worst',worst)
