import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_glyph_scene, load_glyph_fixture
import ref_scenes as RS
imgs=load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
ctx=HipContext(atlas_size=1024, device=0)
w,h=3840,2160
rot=float(os.environ.get('ROT', '0'))  # ROT=2: config 11 (every atlas quad a rotated quad)
sc=make_glyph_scene(w,h,imgs,rotation=rot) if rot else make_glyph_scene(w,h,imgs)
for k,v in RS.used_images(sc,imgs).items(): ctx.put_image(k,v)
ctx.render_frame(sc,w,h); ctx.replay(10); ctx.replay(100); st=ctx.frame_stats(); ms=st.ms_total
ctx.profile(20); st=ctx.frame_stats()
print(f"cfg{11 if rot else 4} T10k@4K: draws={st.n_draws} frame={ms*1e3:.1f} us -> {w*h/ms/1e3:.0f} Mpix/s; bin {st.ms_bin*1e3:.1f} comp {st.ms_composite*1e3:.1f}")
