"""cProfile of the C3 evaluation (host side): what runs besides the native KDE call"""
import sys, cProfile, pstats, io
sys.argv = [sys.argv[0], "1e7", "1"]
exec(open("/root/repo/scripts/dev/c3_probe.py").read().split("times = []")[0])
for it in range(3):
    pipe.params.theta23.value = (42.0 + it) * ureg.degree
    pipe.get_outputs()
pr = cProfile.Profile()
pr.enable()
for it in range(20):
    pipe.params.theta23.value = (45.0 + 0.1 * it) * ureg.degree
    pipe.get_outputs()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
