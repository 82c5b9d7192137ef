ls /sys/class/kfd/kfd/proc/ 2>&1 | head
python - <<'PY'
import torch, os, glob, subprocess, sys, time
x = torch.zeros(1<<28, device="cuda")  # 1 GB
torch.cuda.synchronize()
pid = os.getpid()
print("pid", pid, "free/total", torch.cuda.mem_get_info())
print(os.listdir("/sys/class/kfd/kfd/proc") if os.path.isdir("/sys/class/kfd/kfd/proc") else "no kfd proc dir")
for f in glob.glob(f"/sys/class/kfd/kfd/proc/{pid}/*"):
    try:
        print(f, open(f).read().strip()[:100] if os.path.isfile(f) else "<dir>")
    except Exception as e:
        print(f, "ERR", e)
# second process
p = subprocess.Popen([sys.executable, "-c", "import torch,time; y=torch.zeros(1<<29,device='cuda'); torch.cuda.synchronize(); print('child up', flush=True); time.sleep(6)"], stdout=subprocess.PIPE, text=True)
print(p.stdout.readline().strip())
print("with child: free/total", torch.cuda.mem_get_info())
print(os.listdir("/sys/class/kfd/kfd/proc"))
for d in os.listdir("/sys/class/kfd/kfd/proc"):
    for f in glob.glob(f"/sys/class/kfd/kfd/proc/{d}/vram_*"):
        try: print(f, open(f).read().strip())
        except Exception as e: print(f, "ERR", e)
p.wait()
PY
cat /sys/class/kfd/kfd/topology/nodes/*/gpu_id 2>/dev/null | head
rocm-smi --showpids 2>&1 | head -20
