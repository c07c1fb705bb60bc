
template Program() {
   signal input EokAkccw;
   signal input cgdsYlKM;
   signal input cWIKmUqs;
   signal input kzdaWLwA;
   for (var FxDnLjEJ = 7; (FxDnLjEJ<=37); FxDnLjEJ=(FxDnLjEJ+1)) {
      signal output UBqmhdNS;
      UBqmhdNS<==(((cgdsYlKM+cWIKmUqs)*(kzdaWLwA+EokAkccw))+(EokAkccw-EokAkccw));
      var QqqTfdvN = (8682315818494689271-6669520046870740632);
   }
   var IfzNOdXD = 10037329643339398324;
   signal output MHuhirVZ;
   MHuhirVZ<==(((kzdaWLwA+kzdaWLwA)*(cgdsYlKM+cWIKmUqs))+(2331984072086560701+cgdsYlKM));
   IfzNOdXD=(17597007249729726374-6978347450996498130);
   var OtCMliiY[5];
   var gisimvUc[3];
   signal output ULQkpWfu;
   ULQkpWfu<==(((MHuhirVZ-EokAkccw)*(kzdaWLwA-MHuhirVZ))+17957997619425867047);
   IfzNOdXD=7209496026795652486;
   IfzNOdXD=cWIKmUqs;
   for (var qpiwObSv = 25; (qpiwObSv<=34); qpiwObSv=(qpiwObSv+1)) {
      for (var EUAlRtZX = 5; (EUAlRtZX<=55); EUAlRtZX=(EUAlRtZX+1)) {
         qpiwObSv=3385764088662349309;
         var xZUnqzBr[3];
         xZUnqzBr[2]=18227835141270345680;
      }
      IfzNOdXD=IfzNOdXD;
      for (var ShnsgSem = 2; (ShnsgSem<=32); ShnsgSem=(ShnsgSem+1)) {
         ShnsgSem=(7156081718951498730-1410592953433330230);
         IfzNOdXD=(182638525829597345*9198673184403805956);
      }
   }
   for (var tDuFbxkg = 17; (tDuFbxkg<=41); tDuFbxkg=(tDuFbxkg+1)) {
      signal output tYyIfZnR;
      tYyIfZnR<==(((cWIKmUqs+cgdsYlKM)*(MHuhirVZ+cWIKmUqs))+kzdaWLwA);
   }
   IfzNOdXD=(5566530113884357838+7549872655265273417);
   IfzNOdXD=(1239858701542591727-13266646950593582766);
   signal output hEuhnCOu;
   hEuhnCOu<==(((cgdsYlKM-kzdaWLwA)*(kzdaWLwA+EokAkccw))+(cWIKmUqs+cgdsYlKM));
   OtCMliiY[3]=(10474254879984423835*2297494973244885645);
   IfzNOdXD=(OtCMliiY[3]);
   OtCMliiY[4]=(5599570169126193183-10320950992019848268);
   gisimvUc[1]=(11662849115667298850-9536576354136341929);
   var mjGjmzes;
   mjGjmzes=kzdaWLwA;
   signal output iCOSMszf;
   iCOSMszf<==((5749081643253271876*kzdaWLwA)+(ULQkpWfu+cWIKmUqs));
   signal output ESHUOIyt;
   ESHUOIyt<==((kzdaWLwA*hEuhnCOu)+(cWIKmUqs+ULQkpWfu));
   var ApHKeNhd[9];
   gisimvUc[0]=(OtCMliiY[4]);
   for (var AuXSfoHU = 27; (AuXSfoHU<=46); AuXSfoHU=(AuXSfoHU+1)) {
      for (var bQfFdgll = 21; (bQfFdgll<=42); bQfFdgll=(bQfFdgll+1)) {
         gisimvUc[0]=13936108547183410715;
      }
      signal output rGIKVUaO;
      rGIKVUaO<==(((ULQkpWfu+ULQkpWfu)*(hEuhnCOu-ESHUOIyt))+6212691550337344053);
      mjGjmzes=(gisimvUc[1]);
      var hITNecXC[3];
   }
   for (var NqvHeCHm = 1; (NqvHeCHm<=28); NqvHeCHm=(NqvHeCHm+1)) {
      var yjKeWlwE[9];
   }
   signal output ToVmFiwE;
   ToVmFiwE<==(((10872470939614765455-ULQkpWfu)*(MHuhirVZ-1976518104862652214))+hEuhnCOu);
   for (var YcheHZvQ = 4; (YcheHZvQ<=7); YcheHZvQ=(YcheHZvQ+1)) {
      signal output mGVghGvI;
      mGVghGvI<==((ToVmFiwE*(EokAkccw-EokAkccw))+(ESHUOIyt-5489475628114300457));
      YcheHZvQ=(13021326243499149111-4496986975685146458);
      mjGjmzes=13359183867166702774;
   }
   var okSeVEID[6];
   for (var bivWkcvK = 18; (bivWkcvK<=46); bivWkcvK=(bivWkcvK+1)) {
      var IZGjtKSk;
      var EPPJmTmi[10];
      for (var TrvuSuGa = 30; (TrvuSuGa<=60); TrvuSuGa=(TrvuSuGa+1)) {
         signal output SWXpgyUR;
         SWXpgyUR<==(((ESHUOIyt-1088201090370049925)*(1883454189237844898-ESHUOIyt))+(13913522631409014480+18307576812257953670));
         signal output TLRzZXTd;
         TLRzZXTd<==(((EokAkccw-MHuhirVZ)*(cgdsYlKM+ToVmFiwE))+(cWIKmUqs+kzdaWLwA));
         IfzNOdXD=hEuhnCOu;
      }
   }
   IfzNOdXD=iCOSMszf;
   for (var AOyOWIBk = 13; (AOyOWIBk<=56); AOyOWIBk=(AOyOWIBk+1)) {
      signal output SqHnHohF;
      SqHnHohF<==((ESHUOIyt*(ULQkpWfu-cgdsYlKM))+(kzdaWLwA+1901499401055009892));
      gisimvUc[0]=(13602175448108644176-10124570710261400226);
      var ykjFiBBL[9];
   }
   ApHKeNhd[4]=581563635395940212;
   for (var zAQOYGiU = 10; (zAQOYGiU<=46); zAQOYGiU=(zAQOYGiU+1)) {
      IfzNOdXD=MHuhirVZ;
      signal output atropcDI;
      atropcDI<==((ULQkpWfu*(ESHUOIyt-kzdaWLwA))+cWIKmUqs);
      for (var SEZeGOOK = 5; (SEZeGOOK<=12); SEZeGOOK=(SEZeGOOK+1)) {
         signal output jCQGPWPc;
         jCQGPWPc<==((ULQkpWfu*(ToVmFiwE-ToVmFiwE))+(kzdaWLwA+5626568065001686547));
         var zheDHcFa = 16605174391901886106;
      }
      var vzbBPAvC[4];
      vzbBPAvC[1]=(OtCMliiY[3]);
   }
   signal output EDzZVSiQ;
   EDzZVSiQ<==(((cWIKmUqs+cWIKmUqs)*(ULQkpWfu+iCOSMszf))+ToVmFiwE);
   gisimvUc[2]=cWIKmUqs;
   var DDkungrH = (1297874394224555470+13902978905631996408);
   okSeVEID[5]=hEuhnCOu;
   var mZdzHAIz[10];
   ApHKeNhd[8]=(3996484550252845626*9870357528539203469);
   var VUFuYfhT[10];
   signal output LjscTytc;
   LjscTytc<==(((ToVmFiwE-9610586505108554365)*(ULQkpWfu+ULQkpWfu))+(17142546153766769817+ULQkpWfu));
   var bBDsLbWO = (16551504728630064805*16584305626474837463);
   signal output smfTqGbz;
   smfTqGbz<==(((iCOSMszf-kzdaWLwA)*(iCOSMszf+8908403981786629443))+(EokAkccw-cgdsYlKM));
   var uCBafZTU = (11535607409126280236+706755829666147397);
   bBDsLbWO=kzdaWLwA;
   signal output ghKIVIcR;
   ghKIVIcR<==((cWIKmUqs*smfTqGbz)+LjscTytc);
   var pLQhezrt[10];
   var KSNlTERV = (ApHKeNhd[4]);
   signal output EhFkwNTX;
   EhFkwNTX<==((ghKIVIcR*(iCOSMszf+MHuhirVZ))+(ghKIVIcR-EDzZVSiQ));
   signal output dZMJMYCN;
   dZMJMYCN<==((11322568176538136997*(kzdaWLwA-ULQkpWfu))+(EokAkccw+hEuhnCOu));
   var NTWyvCpE[9];
   signal output jhoQrQWN;
   jhoQrQWN<==((hEuhnCOu*(iCOSMszf+MHuhirVZ))+(7576891972182799923-16079949016652609818));
   uCBafZTU=(1812254395269356394*11676395669394737801);
   mZdzHAIz[2]=ULQkpWfu;
   for (var oBaywjKO = 16; (oBaywjKO<=52); oBaywjKO=(oBaywjKO+1)) {
      IfzNOdXD=(16429097976821377993-16178140865070360886);
      var oFboyVMB[2];
      for (var TvBSfOxy = 28; (TvBSfOxy<=33); TvBSfOxy=(TvBSfOxy+1)) {
         oFboyVMB[0]=14374339925601366772;
      }
      var rkaTyHlm = (ApHKeNhd[8]);
   }
   KSNlTERV=kzdaWLwA;
   for (var wStMKqJf = 3; (wStMKqJf<=20); wStMKqJf=(wStMKqJf+1)) {
      gisimvUc[0]=3045788724628932305;
      signal output zOnfchBO;
      zOnfchBO<==((18007451882401889058*jhoQrQWN)+(dZMJMYCN+cgdsYlKM));
      var MrUtMDvy[7];
      uCBafZTU=18040717821257204189;
   }
   OtCMliiY[4]=1222141343192154061;
   gisimvUc[0]=(9562196185063223166-8424615425965355053);
}

component main = Program();
        