#!/usr/bin/env python
"""Randomised check of pgw4era5_amd/ncio.py against scipy.io.netcdf_file (CPU only): random NetCDF-3 files (classic and
64-bit offset, record and fixed variables of every classic type incl. NC_CHAR, scalars, attributes) written by scipy are
read by the native reader, written again by the native writer and read back by scipy; names, dimensions, dtypes, values
and attributes must survive.  usage: python tools/fuzz_ncio.py [seed] [cases]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scipy.io import netcdf_file
from pgw4era5_amd import ncio


def main(seed=0, cases=300):
    rng=np.random.default_rng(seed)
    tmp=tempfile.mkdtemp(prefix='ncfuzz')
    TYPES=['b','h','i','f','d','c']
    fails=0
    for case in range(cases):
        p=os.path.join(tmp,'c.nc')
        version=int(rng.choice([1,2]))
        nc=netcdf_file(p,'w',version=version)
        nd=int(rng.integers(1,5))
        has_rec=rng.random()<0.6
        dims=[]
        if has_rec:
            nc.createDimension('time',None); dims.append(('time',int(rng.integers(0,5))))
        for k in range(nd):
            n=int(rng.integers(1,6)); nc.createDimension('d%d'%k,n); dims.append(('d%d'%k,n))
        if rng.random()<0.7: nc.title='t%d'%case
        if rng.random()<0.5: nc.fl=np.float32(rng.normal())
        if rng.random()<0.5: nc.arr=rng.integers(-5,5,size=int(rng.integers(1,4))).astype(np.int32)
        if rng.random()<0.3: nc.dbl=np.array(rng.normal(size=2))
        if rng.random()<0.3: nc.sh=np.int16(7)
        nv=int(rng.integers(1,6))
        data={}
        nrec=dims[0][1] if has_rec else None
        for v in range(nv):
            tc=str(rng.choice(TYPES))
            k=int(rng.integers(0,min(4,len(dims))+1))
            pick=sorted(rng.choice(len(dims),size=k,replace=False).tolist()) if k else []
            if has_rec and 0 in pick: pick=[0]+[i for i in pick if i!=0]
            vd=tuple(dims[i][0] for i in pick)
            shape=tuple(dims[i][1] for i in pick)
            var=nc.createVariable('v%d'%v,tc,vd)
            if tc=='c':
                arr=rng.integers(97,123,size=shape).astype('u1').view('S1').reshape(shape) if shape else np.array(b'x',dtype='S1')
            elif tc in 'bhi':
                arr=rng.integers(-100,100,size=shape).astype({'b':'i1','h':'i2','i':'i4'}[tc])
            else:
                arr=rng.normal(size=shape).astype({'f':'f4','d':'f8'}[tc])
            if shape==():
                pass   # scipy cannot assign 0-d values here; it writes its initial value
            elif 0 not in shape or (has_rec and pick and pick[0]==0 and nrec>0 and 0 not in shape[1:]):
                var[:]=arr
            if rng.random()<0.5: var.units='K'
            if rng.random()<0.3 and tc in 'fd': var.scale_factor=np.float64(0.5)
            data['v%d'%v]=(arr,vd)
        try:
            nc.close()
        except Exception as e:
            continue
        try:
            ref=netcdf_file(p,'r',mmap=False)
            mine=ncio.open_dataset(p,decode_times=False,decode_mask_scale=False)
            assert list(mine.variables)==list(ref.variables.keys()),(list(mine.variables),list(ref.variables.keys()))
            for k,rv in ref.variables.items():
                a=np.asarray(rv.data); b=mine[k].values
                assert tuple(mine[k].dims)==tuple(rv.dimensions),(k,mine[k].dims,rv.dimensions)
                assert a.shape==b.shape,(k,a.shape,b.shape)
                assert a.dtype.kind==b.dtype.kind and a.dtype.itemsize==b.dtype.itemsize,(k,a.dtype,b.dtype)
                assert np.array_equal(a,b,equal_nan=(a.dtype.kind=='f')),(k,)
                for an,av in rv._attributes.items():
                    mv=mine[k].attrs[an]
                    if isinstance(av,bytes): assert mv==av.decode() or mv==av,(an,mv,av)
                    else: assert np.array_equal(np.asarray(mv),np.asarray(av)),(an,mv,av)
            for an,av in ref._attributes.items():
                mv=mine.attrs[an]
                if isinstance(av,bytes): assert mv==av.decode() or mv==av,(an,mv,av)
                else: assert np.array_equal(np.asarray(mv),np.asarray(av)),(an,mv,av)
            # round trip through our writer, read back with scipy
            q=os.path.join(tmp,'o.nc')
            ncio.to_netcdf(mine,q)
            back=netcdf_file(q,'r',mmap=False)
            assert list(back.variables.keys())==list(ref.variables.keys())
            for k,rv in ref.variables.items():
                a=np.asarray(rv.data); bv=back.variables[k]
                b=np.asarray(bv.data)
                assert a.shape==b.shape and a.dtype==b.dtype and np.array_equal(a,b,equal_nan=(a.dtype.kind=='f')),(k,a.shape,b.shape,a.dtype,b.dtype)
                assert tuple(bv.dimensions)==tuple(rv.dimensions)
                assert set(bv._attributes)==set(rv._attributes),(k,bv._attributes,rv._attributes)
            assert set(back._attributes)==set(ref._attributes)
            used=any('time' in rv.dimensions for rv in ref.variables.values())
            assert (not used) or ((back.dimensions.get('time','x') is None)==(ref.dimensions.get('time','x') is None))
            ref.close(); back.close()
        except Exception as e:
            fails+=1
            print('case',case,'version',version,'dims',dims,'vars',{k:(v[0].dtype.str,v[1]) for k,v in data.items()},'->',type(e).__name__,str(e)[:300])
            if fails>8: break
    print('fails',fails)
    return fails


if __name__ == '__main__':
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300) else 0)
