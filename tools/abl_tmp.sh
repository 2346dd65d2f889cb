for v in base small; do echo == $v; for i in 1 2 3; do VLGAE_AMD_LIB=$PWD/tools/variants/lib_$v.so python tools/time_headline.py 2>&1 | grep fused; done; done
