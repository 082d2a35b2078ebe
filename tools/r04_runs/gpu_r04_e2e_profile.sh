# where the wall time of an at-size validate() goes: tools/probe_e2e.py under cProfile (host side), sorted by cumulative time
python tools/probe_e2e.py 2>&1 | tail -3
python -c "
import cProfile, pstats, runpy, io, sys
pr = cProfile.Profile(); pr.enable()
runpy.run_path('tools/probe_e2e.py', run_name='__main__')
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
" 2>&1 | cut -c1-170 | tail -75
