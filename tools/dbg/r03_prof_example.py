import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo')
sys.argv = ['x', '10']
import runpy
# warm
src = open('/root/repo/tools/mcp_example_shape.py').read().replace("mm_states=True, mm_rewards=True", "mm_states=False, mm_rewards=False")
code = compile(src, 'ex', 'exec')
g = {'__name__': '__main__', '__file__': '/root/repo/tools/mcp_example_shape.py'}
exec(code, g)
pm, x0, dyn, pol, opt, torch = g['pm'], g['x0'], g['dyn'], g['pol'], g['opt'], g['torch']
pr = cProfile.Profile()
pr.enable()
pm.algorithms.mc_pilco(x0, dyn, pol, 15, opt, None, 300)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:5000])
