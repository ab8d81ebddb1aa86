"""Re-print julia/KlaraHIP/src/KlaraHIP.jl inside INTEGRATION.md (tests/test_host_api.py checks the block is verbatim)."""
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
jl = (ROOT / "julia" / "KlaraHIP" / "src" / "KlaraHIP.jl").read_text()
md = (ROOT / "INTEGRATION.md").read_text()
a = md.index("```julia\n# KlaraHIP.jl")
b = md.index("end # module\n```", a) + len("end # module\n```")
(ROOT / "INTEGRATION.md").write_text(md[:a] + "```julia\n" + jl + "```" + md[b:])
