function y = invHHt(v,p,q,sg,phi)
% Drop-in shim with the reference's signature (Class2/invHHt.m:1); forwards to libipdamg.
y = ipd_mex('invHHt', v, p, q, sg, phi);
end
