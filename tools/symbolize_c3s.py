"""Attribution of the stack frames of gpurun_out/c3s.log (round 2 SIGSEGV under rocprofv3) to libraries of this image without\nsymbols or /proc/maps: a frame is a return address, so it must be preceded by a call instruction; for a group of frames of one\nlibrary only one page-aligned load base satisfies that for all of them."""
import subprocess, sys, struct, bisect, re
libs = {
 'libamdhip64': '/opt/rocm/lib/libamdhip64.so.7.2.70200',
 'libhsa-runtime64': '/opt/rocm/lib/libhsa-runtime64.so.1.18.70200',
 'librocprofiler-sdk': '/opt/rocm/lib/librocprofiler-sdk.so.1.1.0',
 'librocprofiler-sdk-tool': '/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so.1.1.0',
 'libc': '/lib/x86_64-linux-gnu/libc.so.6',
 'libstdc++': '/lib/x86_64-linux-gnu/libstdc++.so.6',
 'torch_hip': None,
}
groups = {
 'A(pc)': [0x7022ab7412fb],
 'B': [0x7022a0c58266, 0x7022a0c495c0],
 'C': [0x7021e6bb6abe, 0x7021e6bb2fac, 0x7021e6bb3684, 0x7021e6b7c1d6, 0x7021e6a18ca4, 0x7021e6a68782, 0x7021e6a1931e, 0x7021e6a35206],
 'D': [0x7022abaa1ec0],
 'sig': [0x7022ab28dee8, 0x7022ac15250e, 0x7022ab236520],
}
def text_of(path):
    out = subprocess.run(['readelf','-S','-W',path],capture_output=True,text=True).stdout
    secs=[]
    for l in out.split('\n'):
        m=re.match(r'\s*\[\s*\d+\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)',l)
        if m and m.group(1) in ('.text','.plt','.plt.sec','.plt.got'):
            secs.append((m.group(1),int(m.group(2),16),int(m.group(3),16),int(m.group(4),16)))
    data=open(path,'rb').read()
    return secs,data
def is_ret_addr(secs,data,va):
    for name,addr,off,size in secs:
        if addr<=va<addr+size and name=='.text':
            fo=off+(va-addr)
            b=data[fo-7:fo]
            if b[-5]==0xE8: return True                       # call rel32
            if b[-2]==0xFF and (b[-1]&0x38)==0x10: return True   # call r/m (2 bytes)
            if b[-3]==0xFF and (b[-2]&0x38)==0x10: return True   # call [reg+disp8]
            if b[-6]==0xFF and (b[-5]&0x38)==0x10: return True   # call [reg+disp32]
            if b[-3]==0x41 and b[-2]==0xFF: return True
            if b[-7]==0xFF and (b[-6]&0x38)==0x10: return True
            return False
    return False
def syms(path):
    out = subprocess.run(['nm','-D','-C','--defined-only',path],capture_output=True,text=True).stdout
    out += subprocess.run(['nm','-C','--defined-only',path],capture_output=True,text=True).stdout
    s=[]
    for l in out.split('\n'):
        p=l.split(' ',2)
        if len(p)==3 and p[1] in 'TtWw':
            try: s.append((int(p[0],16),p[2]))
            except: pass
    s=sorted(set(s))
    return s
for lname,path in libs.items():
    if not path: continue
    secs,data=text_of(path)
    tx=[s for s in secs if s[0]=='.text'][0]
    sy=syms(path); keys=[a for a,_ in sy]
    for gname,addrs in groups.items():
        if gname in ('A(pc)','D','sig'): continue
        lo=min(addrs); hi=max(addrs)
        # candidate bases: page aligned, such that lo-base >= text start and hi-base < text end
        bmin=(hi-(tx[1]+tx[3]))&~0xfff; bmax=(lo-tx[1])&~0xfff
        hits=[]
        b=bmin
        while b<=bmax:
            if all(is_ret_addr(secs,data,a-b) for a in addrs): hits.append(b)
            b+=0x1000
        print(lname,gname,'candidate bases',len(hits),[hex(h) for h in hits[:5]])
        for h in hits[:3]:
            for a in addrs:
                va=a-h; i=bisect.bisect_right(keys,va)-1
                print('    ',hex(a),'->',hex(va),sy[i][1][:110] if i>=0 else '?', '+%#x'%(va-sy[i][0]) if i>=0 else '')
