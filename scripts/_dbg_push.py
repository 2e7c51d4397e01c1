import sys, torch
sys.path.insert(0, '/root/repo')
import cusrl_amd as cusrl
from cusrl_amd import ops
cusrl.config.set_device('cuda:0')
cusrl.set_global_seed(0)
env = cusrl.testing.SyntheticEnvironment(4096, 48, 12, device='cuda:0')
tr = cusrl.Trainer(env, cusrl.preset.PpoAgentFactory(), num_iterations=1, verbose=False)
orig = ops.buffer_push
seen = [0]
def dbg(pairs, cursor, N):
    if seen[0] < 2:
        for (step, storage), name in zip(pairs, tr.agent.buffer.storage):
            nb = step.numel()*step.element_size()
            print(name, tuple(step.shape), step.dtype, 'src%16', step.data_ptr()%16, 'dst%16', (storage.data_ptr()+cursor*nb)%16, 'bytes%16', nb%16, 'contig', step.is_contiguous())
        seen[0]+=1
    return orig(pairs, cursor, N)
ops.buffer_push = dbg
tr.run_training_loop()
