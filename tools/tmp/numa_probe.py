import os, sys, glob
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
p = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
print('gpu0', bdf)
try:
    print('numa_node', open('/sys/bus/pci/devices/%s/numa_node' % bdf).read().strip())
except Exception as e:
    print('numa_node err', e)
for n in sorted(glob.glob('/sys/devices/system/node/node*')):
    try:
        print(n, open(n + '/cpulist').read().strip())
    except Exception as e:
        print(n, e)
print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], '...')
print('cpu_count', os.cpu_count())
os.system('cat /proc/meminfo | head -3; nproc; lscpu | grep -i -E "numa|socket|model name" | head')
