import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100, make_glyph_scene, load_glyph_fixture
import ref_scenes as RS
def run(name, ctx, sc, w, h, n=100):
    t=time.perf_counter(); ctx.render_frame(sc, w, h); ctx.sync(); t1=time.perf_counter()-t
    ctx.replay(10); ctx.replay(n); st=ctx.frame_stats(); ms=st.ms_total
    ctx.profile(20); st=ctx.frame_stats()
    print(f"{name}: {w}x{h} draws={st.n_draws} phases={st.n_phases} frame={ms*1e3:.1f} us -> {w*h/ms/1e3:.0f} Mpix/s; first-frame host {t1*1e3:.1f} ms; bin {st.ms_bin*1e3:.1f} comp {st.ms_composite*1e3:.1f} blur {1e3*(st.ms_blur_h+st.ms_blur_v):.1f} us")
ctx=HipContext(device=0)
run('cfg1 rgb_boxes_sdf', ctx, RS.rgb_boxes_sdf(800.,600.), 800, 600)
run('cfg2 S100@1080p', ctx, make_render_tree_100(1920,1080,0), 1920, 1080)
run('cfg3 S300@4K', ctx, make_render_tree_100(3840,2160,0,full_frame_blur=True), 3840, 2160)
run('cfg5 S300@8K frame', ctx, make_render_tree_100(7680,4320,0,full_frame_blur=True), 7680, 4320, 30)
ctx.close()
imgs=load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
ctx=HipContext(atlas_size=1024, device=0)
sc=make_glyph_scene(3840,2160,imgs)
for k,v in RS.used_images(sc,imgs).items(): ctx.put_image(k,v)
run('cfg4 T10k@4K', ctx, sc, 3840, 2160)
