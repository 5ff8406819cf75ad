import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA_ROOT = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(DATA_ROOT, "data")
