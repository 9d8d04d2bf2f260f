"""bench.py on another build of libcnk (CNK_LIB_OVERRIDE=path): A/B of debug / variant builds at step level."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convnets_amd import _lib
_lib.LIB_PATH = os.environ['CNK_LIB_OVERRIDE']
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
