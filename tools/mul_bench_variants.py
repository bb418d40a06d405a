import pbc_amd
for w in (1,2):
    for v in (6,7,8,9,10):
        r,ms=pbc_amd.mul_bench(v, 4000, w)
        print("waves",w,"variant",v,"%.3f G products/s"%(r/1e9), "%.2f ms"%ms)
