import effort_amd as ea, effort_amd._lib as L
g = ea.gpu(0)
for W,E in [(16,2),(16,1),(8,2),(8,1),(8,4),(4,2),(4,4)]:
    for lds in (20000, 36000, 70000):
        print(W,E,lds, L.lib().effort_debug_occupancy(g.ctx, 0, W, E, lds))
