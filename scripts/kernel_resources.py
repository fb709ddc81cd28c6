"""Registers, scratch and occupancy of every kernel of the product library, from the compiler's own report
(ecog2txt_amd/csrc/build/*.log: build.sh compiles with -Rpass-analysis=kernel-resource-usage).  usage: kernel_resources.py [regex]"""
import re,sys,glob,subprocess,os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sorted(glob.glob(os.path.join(ROOT,'ecog2txt_amd','csrc','build','*.log'))):
    cur={}
    for line in open(f):
        m=re.search(r'remark: (?:[^:]*:\d+:\d+: )?\s*(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\S+)',line)
        if not m:
            m=re.search(r'(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)',line)
        if not m: continue
        k,v=m.group(1),m.group(2)
        if k in('Function Name','Name'):
            cur={'name':v}
        else:
            cur[k]=v
            if k.startswith('LDS'):
                n=subprocess.run(['/usr/bin/c++filt',cur['name']],capture_output=True,text=True).stdout.strip()
                if len(sys.argv)>1 and not re.search(sys.argv[1],n): continue
                print(f"{n[:70]:70s} V{cur.get('VGPRs'):>4} A{cur.get('AGPRs'):>4} S{cur.get('TotalSGPRs'):>4} occ{cur.get('Occupancy [waves/SIMD]'):>2} scr{cur.get('ScratchSize [bytes/lane]'):>4} lds{cur.get('LDS Size [bytes/block]')}")
