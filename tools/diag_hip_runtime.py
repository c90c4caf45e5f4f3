import sys, os
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'libhsa-runtime' in l})
if order == "lib_first":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import turbo_amd as ta
    g = ta.NativeGP(0, "f64")
    print("after lib:", maps())
    import torch
    print("after import torch:", maps())
    try:
        print("torch cuda avail:", torch.cuda.is_available(), torch.cuda.device_count())
        x = torch.zeros(4, device="cuda:0"); print("ok", x.sum().item())
    except Exception as e:
        print("torch failed:", e)
else:
    import torch
    print("torch cuda avail:", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda:0")
    print("after torch:", maps())
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import turbo_amd as ta
    g = ta.NativeGP(0, "f64")
    print("after lib:", maps())
